#!/usr/bin/env python3
"""csplat_gnn_edge_mlp3 (one launch) against the three csplat_linear128 launches it replaces, E = 300,000 (config 4's edge count):
kernel time by HIP events over 50 calls each, alternated."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
from meshnet.graph_ops import absmax, edge_mlp3, edge_mlp3_mode, edge_mlp3_pack, linear128  # noqa: E402

E, N = int(os.environ.get("E", 300_000)), 10_000
gen = torch.Generator().manual_seed(0)
e0 = torch.randn(E, 128, generator=gen).cuda()
W = [(torch.randn(128, 128, generator=gen) * 0.1).cuda() for _ in range(3)]
b = [torch.randn(128, generator=gen).cuda() for _ in range(3)]
xa, xb = torch.randn(N, 128, generator=gen).cuda(), torch.randn(N, 128, generator=gen).cuda()
# config 4's graph shape: 30 neighbours of a node are nearby nodes
dst = torch.arange(N).repeat_interleave(E // N)[:E]
src = (dst + torch.randint(-60, 60, (E,), generator=gen)).clamp(0, N - 1)
perm = torch.argsort(src * N + dst)
ia, ib = dst[perm].cuda(), src[perm].cuda()
norm = torch.nn.LayerNorm(128).cuda()
edge_mlp3_mode(int(os.environ.get("EM_MODE", "0")))      # 0: two fp16 pieces, 1: three bf16 pieces
img = edge_mlp3_pack(*W)
amax = absmax(e0)
out = torch.empty_like(e0)
t1, t2 = torch.empty_like(e0), torch.empty_like(e0)


AGG = bool(int(os.environ.get("EM_AGG", "0")))      # 1: the launch sums its messages per destination (edges in destination order, pieces out)
if AGG:
    from meshnet.graph_ops import GraphCSR, gather_rows
    plan = GraphCSR(torch.stack([ib, ia]), N).agg_plan()
    e0p = gather_rows(e0, plan["perm"])
    pieces = torch.empty(plan["npieces"], 128, device="cuda")


def fused():
    if AGG:
        edge_mlp3(e0p, 4.0, xa, plan["dst"], xb, plan["src"], img, b[0], b[1], b[2], norm, e0_absmax=amax, agg=(plan["gp0"], pieces))
    else:
        edge_mlp3(e0, 4.0, xa, ia, xb, ib, img, b[0], b[1], b[2], norm, out=out, e0_absmax=amax)


def three():
    linear128(e0, W[0], b[0], alpha=4.0, relu=True, gather=(xa, ia, xb, ib), out=t1)
    linear128(t1, W[1], b[1], relu=True, out=t1)
    linear128(t1, W[2], b[2], layer_norm=norm, out=t2)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    bb.record()
    torch.cuda.synchronize()
    return a.elapsed_time(bb) / n * 1e3


with torch.no_grad():
    for rep in range(3):
        print(f"E={E}: one launch {timed(fused):7.1f} us   three launches {timed(three):7.1f} us", flush=True)
    if not AGG:
        print("max |diff| / scale:", float((out - t2).abs().max() / t2.abs().max()))
