#!/usr/bin/env python3
"""tools/host_micro.py -- host microseconds of the Python idioms the product's wrappers use around a C-ABI call (GPU box): what a
launch costs before the library is entered"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from csplat import native as n
dev = torch.device("cuda:0")
x = torch.zeros(1000, 3, device=dev)
N = 20000


def bench(name, fn):
    for _ in range(200): fn()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    print("%-44s %6.2f us" % (name, (time.perf_counter() - t0) / N * 1e6))


def ctx():
    with torch.cuda.device(dev): pass
bench("with torch.cuda.device(dev)", ctx)
bench("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
bench("torch.cuda.current_device()", lambda: torch.cuda.current_device())
bench("torch.empty(1000,3)", lambda: torch.empty(1000, 3, dtype=torch.float32, device=dev))
bench("torch.empty_like", lambda: torch.empty_like(x))
bench("x.data_ptr()", lambda: x.data_ptr())
bench("n.ptr(x)", lambda: n.ptr(x))
bench("x.contiguous()", lambda: x.contiguous())
bench("x.reshape(-1)", lambda: x.reshape(-1))
bench("x[0]", lambda: x[0])
bench("ctypes call (csplat_debug_flags)", lambda: n.lib.csplat_debug_flags(0))
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a
    @staticmethod
    def backward(ctx, g): return g
xr = x.clone().requires_grad_()
bench("Function.apply (identity, requires grad)", lambda: F.apply(xr))
bench("x.shape[0]; int()", lambda: int(x.shape[0]))
bench("x.is_cuda and x.dtype == f32", lambda: x.is_cuda and x.dtype == torch.float32)
s = torch.cuda.Stream()
bench("stream_handle()", lambda: n.stream_handle(dev))
y = torch.zeros(())
bench("float() of a CPU 0-dim", lambda: float(y))
g = torch.ones((), device=dev)
bench("k launch: x.add_(1) (torch op)", lambda: x.add_(1))
torch.cuda.synchronize()
