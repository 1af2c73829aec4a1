#!/usr/bin/env python3
"""Where a wave of csplat_gnn_edge_mlp3 spends a round: s_memtime stamps left by wave 0 of every workgroup at the phase boundaries
(csplat_debug_stamps).  GPU box: python3 tools/edge_mlp3_stamps.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
from csplat import native  # noqa: E402
from meshnet.graph_ops import absmax, edge_mlp3, edge_mlp3_mode, edge_mlp3_pack  # noqa: E402

E, N = 300_000, 10_000
gen = torch.Generator().manual_seed(0)
e0 = torch.randn(E, 128, generator=gen).cuda()
W = [(torch.randn(128, 128, generator=gen) * 0.1).cuda() for _ in range(3)]
b = [torch.randn(128, generator=gen).cuda() for _ in range(3)]
xa, xb = torch.randn(N, 128, generator=gen).cuda(), torch.randn(N, 128, generator=gen).cuda()
dst = torch.arange(N).repeat_interleave(E // N)[:E]
src = (dst + torch.randint(-60, 60, (E,), generator=gen)).clamp(0, N - 1)
perm = torch.argsort(src * N + dst)
ia, ib = dst[perm].cuda(), src[perm].cuda()
norm = torch.nn.LayerNorm(128).cuda()
edge_mlp3_mode(int(os.environ.get("EM_MODE", "0")))      # 0: two fp16 pieces, 1: three bf16 pieces
img = edge_mlp3_pack(*W)
amax = absmax(e0)
out = torch.empty_like(e0)
AGG = bool(int(os.environ.get("EM_AGG", "0")))      # 1: the launch sums its messages per destination (what the rollout runs)
if AGG:
    from meshnet.graph_ops import GraphCSR, gather_rows
    plan = GraphCSR(torch.stack([ib, ia]), N).agg_plan()
    e0p = gather_rows(e0, plan["perm"])
    pieces = torch.empty(plan["npieces"], 128, device="cuda")
buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3):
        (edge_mlp3(e0p, 4.0, xa, plan["dst"], xb, plan["src"], img, b[0], b[1], b[2], norm, e0_absmax=amax, agg=(plan["gp0"], pieces)) if AGG else
         edge_mlp3(e0, 4.0, xa, ia, xb, ib, img, b[0], b[1], b[2], norm, out=out, e0_absmax=amax))
    torch.cuda.synchronize()
    native.lib.csplat_debug_stamps(buf.data_ptr(), buf.numel() * 8)
    (edge_mlp3(e0p, 4.0, xa, plan["dst"], xb, plan["src"], img, b[0], b[1], b[2], norm, e0_absmax=amax, agg=(plan["gp0"], pieces)) if AGG else
         edge_mlp3(e0, 4.0, xa, ia, xb, ib, img, b[0], b[1], b[2], norm, out=out, e0_absmax=amax))
    torch.cuda.synchronize()
    native.lib.csplat_debug_stamps(None, 0)
s = buf.cpu().numpy().reshape(256, 64).astype(np.int64)
if False:
    pass
else:       # k_edge_mlp3r (weights in registers): a pair of 32-row tiles, one phase = the 24 (fp16 pieces) / 48 (bf16 pieces) MFMAs of one tile's layer + the side work in its gaps + barrier
    names = ["0: A layer 1 | LN partials B', LN end + rows out A'", "1: B layer 1 | ReLU A, LN end + rows out B'", "2: A layer 2 | ReLU B, gathers A+",
             "3: B layer 2 | ReLU A, gathers B+", "4: A layer 3 | ReLU B, edge rows A+", "5: B layer 3 | LN partials A, edge rows B+"]
    unit = "pair of 32-row tiles"
NS = len(names) + 1       # stamps per round incl. the next round's first
for r in range(3 if NS == 13 else 8):
    seg = s[:, (NS - 1) * r:(NS - 1) * r + NS]
    ok = (seg > 0).all(1)
    d = np.diff(seg[ok], axis=1)
    print(f"{unit} {r}: {int(ok.sum())} workgroups, total {d[:, :NS - 1].sum(1).mean():.0f} cycles")
    for k in range(NS - 1):
        print(f"    {names[k]:48s} mean {d[:, k].mean():8.0f}  median {np.median(d[:, k]):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
