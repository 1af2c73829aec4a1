"""tools/k6_stats.py -- what the compositing kernels walk on scene_1 (one view, P = 100k, 800x800): tile-list entries, (entry, block)
pairs that REACH a 4x4 block (K5b's masks), pairs a block BLENDED (K6's bbits), survivors per (segment, block), steps of 16 and groups of
4 incl. padding.  GPU box."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import util
from csplat import synthetic as syn

P, W, H = 100_000, 800, 800
sc = syn.scene_1(P=P, W=W, H=H, n_cams=1)
case = dict(g=syn.gaussians_at(sc), cam=sc["cameras"][0], W=W, H=H, P=P, bg=sc["bg"], sh_degree=3)
color, radii, depth, st = util.gpu_forward_raw(case)
R = st["R"]
import diff_gaussian_rasterization as dgr  # noqa
# raw binning chunk: recompute the layout (csplat_raster.hip: binning_offsets)
a256 = lambda x: (x + 255) // 256 * 256
tiles = ((W + 15) // 16) * ((H + 15) // 16)
SEG = 256
n = max(R, 1)
slots = R // SEG + tiles + 1
off = [0]
off.append(a256(n * 8)); off.append(off[1] + a256(n * 4)); off.append(off[2] + a256((tiles + 1) * 4 + tiles * 16 * 4))
off.append(off[3] + a256(slots * 4)); off.append(off[4] + a256(slots * 256 * 16)); off.append(off[5] + a256((n + 1) * 2))
off.append(off[6] + a256((n + 1) * 16)); off.append(off[7] + a256((n + 1) * 16)); off.append(off[8] + a256((n + 1) * 8))
raw = st["_binning_raw"] if "_binning_raw" in st else None
if raw is None:
    raise SystemExit("util.gpu_forward_raw does not expose the raw binning chunk (st['_binning_raw'])")
raw = raw.cpu().numpy()
mask16 = raw[off[5]:off[5] + 2 * R].view(np.uint16)
seg_off = raw[off[2]:off[2] + 4 * (tiles + 1)].view(np.int32)
nslots = int(seg_off[tiles])
bb = raw[off[9]:off[9] + nslots * 16 * 32].view(np.uint64).reshape(nslots, 16, 4)
blk_hi = raw[off[2] + 4 * (tiles + 1):off[2] + 4 * (tiles + 1) + tiles * 64].view(np.uint32).reshape(tiles, 16)
popc16 = np.array([bin(i).count("1") for i in range(65536)], np.int64)
reach = int(popc16[mask16].sum())
ranges = st["ranges"]
# live (slot, block) pairs = those K7 launches a wave for: blk_hi > seg_lo
live = 0; blended = 0; groups4 = 0; steps16_reach = 0; surv_hist = []; wave_steps = []
for t in range(tiles):
    n_t = int(ranges[t, 1] - ranges[t, 0])
    if n_t == 0:
        continue
    s0 = int(seg_off[t]); ns = (n_t + SEG - 1) // SEG
    m = mask16[ranges[t, 0]:ranges[t, 1]]
    for b in range(16):
        bits = (m >> b) & 1
        wave_steps.append(0)
        for s in range(ns):
            r_ = int(bits[s * SEG:(s + 1) * SEG].sum())
            if s * SEG < blk_hi[t, b]:
                live += 1
                w = bb[s0 + s, b]
                c = sum(bin(int(x)).count("1") for x in w)
                blended += c; groups4 += (c + 3) // 4
                surv_hist.append(c)
            # K6 walks up to blk_hi (it stops when every pixel is done): count the reach entries in front of blk_hi
            hi = min((s + 1) * SEG, int(blk_hi[t, b]) if blk_hi[t, b] > 0 else 0)
            if hi > s * SEG:
                rr = int(bits[s * SEG:hi].sum())
                steps16_reach += (rr + 15) // 16
                wave_steps[-1] += (rr + 15) // 16
sh = np.array(surv_hist)
print(f"R = {R} entries, {nslots} segments, reach pairs {reach} ({reach / R:.2f} blocks per entry), live (segment, block) pairs {live}")
print(f"blended pairs {blended} ({blended / max(reach, 1):.2f} of reach); K7 groups of 4 (padded) {groups4} = {4 * groups4} slots, fill {blended / max(4 * groups4, 1):.2f}")
print(f"blended survivors per live (segment, block): mean {sh.mean():.1f}, median {np.median(sh):.0f}, p90 {np.percentile(sh, 90):.0f}, zero {float((sh == 0).mean()):.2f}")
print(f"K6 steps of 16 over the reach entries in front of blk_hi: {steps16_reach} = {16 * steps16_reach} slots")
ws = np.array(wave_steps)
print(f"K6 waves (non-empty tiles x 16 blocks): {len(ws)}; steps per wave mean {ws.mean():.1f}, median {np.median(ws):.0f}, p90 {np.percentile(ws, 90):.0f}, "
      f"p99 {np.percentile(ws, 99):.0f}, max {ws.max()}; 5120 wave slots -> {ws.sum() / 5120:.1f} steps per slot if perfectly balanced")
# how many 64-byte atomic requests K7 would issue if the row sums of an entry were joined over larger pixel areas before leaving the CU
bbv = bb[:nslots]                                            # [slot][block][4 words]
def popc(a):
    a = a.copy(); c = np.zeros(a.shape, np.int64)
    while a.any():
        c += (a & np.uint64(1)).astype(np.int64); a >>= np.uint64(1)
    return c
per_block = int(popc(bbv).sum())
pair_h = int(popc(bbv[:, 0::2] | bbv[:, 1::2]).sum())        # blocks (2k, 2k+1): 8x4 pixels
quad = bbv.reshape(nslots, 2, 2, 2, 2, 4)                    # [by/2][by%2][bx/2][bx%2]
quad_u = quad[:, :, 0, :, 0] | quad[:, :, 0, :, 1] | quad[:, :, 1, :, 0] | quad[:, :, 1, :, 1]
tile_u = np.bitwise_or.reduce(bbv, axis=1)
print(f"atomic requests per view if joined per: block {per_block}, horizontal block pair {pair_h}, 8x8 quadrant {int(popc(quad_u).sum())}, "
      f"tile {int(popc(tile_u).sum())}  (list entries {R})")
# K7's workgroups pack a slot's LIVE blocks (blk_hi > seg_lo) four at a time: requests if a workgroup joined its blocks' sums, for two pack orders
qorder = [0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15]
def packed(order):
    tot = 0; wgs = 0
    for t in range(tiles):
        n_t = int(ranges[t, 1] - ranges[t, 0])
        if n_t == 0:
            continue
        s0 = int(seg_off[t]); ns = (n_t + SEG - 1) // SEG
        for s_ in range(ns):
            livebl = [b for b in order if s_ * SEG < blk_hi[t, b]]
            for i in range(0, len(livebl), 4):
                u = np.bitwise_or.reduce(bbv[s0 + s_, livebl[i:i + 4]], axis=0)
                tot += sum(bin(int(x)).count("1") for x in u); wgs += 1
    return tot, wgs
for name, order in (("index order", list(range(16))), ("quadrant-major order", qorder)):
    tot, wgs = packed(order)
    print(f"live blocks packed four to a workgroup in {name}: {wgs} workgroups, {tot} requests per view")
