#!/usr/bin/env python3
"""tools/bench_linear128_train.py [M] -- the csplat_linear128_ex variants of the training path (HIP-event timed, us per call)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from meshnet.graph_ops import linear128, _unit_ln

M = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
dev = "cuda"
A = torch.randn(M, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.1; b = torch.randn(128, device=dev)
Wf = torch.randn(128, 384, device=dev) * 0.1
acc = torch.randn(M, 128, device=dev)
out = torch.empty_like(A)
Wt = W.t().contiguous()
stats = torch.empty(M, 2, device=dev)
N = 10_000
xa, xb = torch.randn(N, 128, device=dev), torch.randn(N, 128, device=dev)
ia = torch.sort(torch.randint(0, N, (M,), device=dev)).values; ib = torch.randint(0, N, (M,), device=dev)

def timed(fn, n=30):
    for _ in range(5): fn()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a_.record()
    for _ in range(n): fn()
    b_.record(); torch.cuda.synchronize()
    return a_.elapsed_time(b_) / n * 1e3

cases = {
    "plain fwd (W, bias, relu)": lambda: linear128(A, W, b, relu=True, out=out),
    "plain, no bias / relu": lambda: linear128(A, W, out=out),
    "W.t() in place": lambda: linear128(A, W.t(), out=out),
    "W.t().contiguous()": lambda: linear128(A, Wt, out=out),
    "W.t() + float mask": lambda: linear128(A, W.t(), mask=acc, out=out),
    "W.t() + add_post": lambda: linear128(A, W.t(), add_post=acc, out=out),
    "column slice": lambda: linear128(A, Wf[:, 256:], out=out),
    "LN epilogue + stats": lambda: linear128(A, W, b, layer_norm=_unit_ln(dev, 1e-5), ln_stats=stats, out=out),
    "gather + relu": lambda: linear128(A, W, None, alpha=2.0, relu=True, gather=(xa, ia, xb, ib), out=out),
}
for name, fn in cases.items():
    print("%-28s %7.1f us" % (name, timed(fn)))
