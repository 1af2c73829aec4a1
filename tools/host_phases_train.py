#!/usr/bin/env python3
"""tools/host_phases_train.py -- host microseconds of the phases of csplat.train.train_step (perf_counter around the calls; the GPU
runs freely): where the config-3 step, which is bound by the host's launch rate, spends its time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from csplat import train as tr, optim as op
import gaussian_renderer as gr
import diff_gaussian_rasterization as dgr

acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: acc.setdefault(name, []).append(time.perf_counter() - t0)
    return w
tr.render_views = timed("render_views", tr.render_views)
tr.regularization = timed("regularization", tr.regularization)
tr.image_losses = timed("image_losses", tr.image_losses)
tr.psnr = timed("psnr", tr.psnr)
op.GroupedAdam.step = timed("GroupedAdam.step", op.GroupedAdam.step)
torch.Tensor.backward = timed("loss.backward", torch.Tensor.backward)
B = dgr._RasterizeGaussiansBatch
B.forward = staticmethod(timed("Batch.forward", B.forward)); B.backward = staticmethod(timed("Batch.backward", B.backward))
gr.rasterize_views = timed("rasterize_views", gr.rasterize_views)
orig_step = tr.train_step
tr.train_step = timed("train_step", orig_step)
sys.argv = ["bench_train.py", "--steps", "200", "--warmup", "20"]
import bench_train
from meshnet import meshnet_network as mm
for cls in (getattr(mm, "ResidualMeshSimulator", None),):
    if cls is not None and hasattr(cls, "forward_times"):
        cls.forward_times = timed("simulator.forward_times", cls.forward_times)
bench_train.main()
for k, v in acc.items():
    v = v[len(v) // 4:]
    print("%-28s n/step %.1f  mean %7.1f us" % (k, len(acc[k]) / len(acc["train_step"]), 1e6 * sum(v) / len(v)))
