#!/usr/bin/env python3
"""Host-side profile of the reference's call pattern (one GaussianRasterizer call per camera, one loss, one backward) at config-2 size:
cProfile over 100 steps, and the step time with PER_CALL_SPECULATION on / off."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
from csplat import synthetic as syn  # noqa: E402
from csplat.train import l1_loss  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = torch.device("cuda:0")
P, W, H, V = 100_000, 800, 800, 4
sc = syn.scene_1(P=P, W=W, H=H, n_cams=V)
g = syn.gaussians_at(sc)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731
params = {k: T(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=T(sc["bg"]), scale_modifier=1.0,
                                          viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                                          campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in sc["cameras"]]
targets = [torch.rand(3, H, W, device=dev) for _ in range(V)]
zeros = torch.zeros(V, P, 3, device=dev)
one = torch.ones((), device=dev)


def step():
    for p in params.values():
        p.grad = None
    m2d = [zeros[i].detach().requires_grad_() for i in range(V)]
    outs = [GaussianRasterizer(settings[i])(means3D=params["means3D"], means2D=m2d[i], opacities=params["opacities"], shs=params["shs"],
                                            scales=params["scales"], rotations=params["rotations"]) for i in range(V)]
    loss = torch.stack([l1_loss(outs[i][0], targets[i]) for i in range(V)]).mean()
    loss.backward(gradient=one)


def timed(n=50):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    for spec in (True, False):
        dgr.PER_CALL_SPECULATION = spec
        print(f"per-call speculation {spec}: {timed():.3f} ms per step", flush=True)
dgr.PER_CALL_SPECULATION = bool(int(os.environ.get("SPEC", "1")))
import gc
gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
