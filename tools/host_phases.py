#!/usr/bin/env python3
"""tools/host_phases.py -- host time of the library calls and of the autograd hand-over inside the bench step (perf_counter around the
ctypes calls; the GPU runs ahead or behind freely): where the step's host-only microseconds go"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from csplat import native as _n
import diff_gaussian_rasterization as dgr

acc = {}
def wrap(obj, name):
    fn = getattr(obj, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc.setdefault(name, []).append(time.perf_counter() - t0)
    setattr(obj, name, timed)

class LibProxy:
    def __init__(self, lib): self.__dict__["_lib"] = lib
    def __getattr__(self, n): return getattr(self._lib, n)
proxy = LibProxy(_n.lib)
for n in ("csplat_forward_views", "csplat_backward_views", "csplat_l1"):
    setattr(proxy, n, getattr(_n.lib, n)); wrap(proxy, n)
_n.lib = proxy
B = dgr._RasterizeGaussiansBatch
for n in ("forward", "backward", "_plan_backward"):
    f = getattr(B, n)
    def mk(f, n):
        def timed(*a, **k):
            t0 = time.perf_counter()
            try: return f(*a, **k)
            finally: acc.setdefault("Batch." + n, []).append(time.perf_counter() - t0)
        return staticmethod(timed)
    setattr(B, n, mk(f, n))
sys.argv = ["bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-train-step"]
exec(open(os.path.join(ROOT, "bench.py")).read())
for k, v in acc.items():
    v = v[len(v) // 2:]
    print("%-28s calls/step %.1f  mean %.1f us" % (k, 1.0, 1e6 * sum(v) / len(v)))
