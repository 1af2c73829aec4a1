#!/usr/bin/env python3
"""tools/bench_dw128.py [M ...] -- csplat_dw128 against the library spellings of g^T @ x, HIP-event timed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from meshnet.graph_ops import dw128

def timed(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M in [int(a) for a in sys.argv[1:]] or [10_000, 300_000]:
    g, x = torch.randn(M, 128, device="cuda"), torch.randn(M, 128, device="cuda")
    t = timed(lambda: dw128(g, x))
    tl = timed(lambda: g.t() @ x)
    print("M %7d  csplat_dw128 %7.1f us (%.0f GB/s, %.1f TFLOP/s)   g.t() @ x %7.1f us" % (M, t, 2 * M * 512 / t / 1e3, 2 * M * 128 * 128 / t / 1e6, tl))
