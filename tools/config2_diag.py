#!/usr/bin/env python3
"""tools/config2_diag.py -- BASELINE config 2 at full size: per-gradient error of (HIP fp32 vs oracle fp64) next to (oracle fp32 vs
oracle fp64), per view, and where the largest deviations sit.  Diagnostic for tests/test_raster_gpu.py::test_config2_full_size_vs_oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import util
from util import oracle_forward
from csplat import native, synthetic as syn
from diff_gaussian_rasterization import rasterize_views

P, W, H, V = 100_000, 800, 800, int(os.environ.get("VIEWS", "2"))
sc = syn.scene_1(P=P, W=W, H=H, n_cams=4)
g = syn.gaussians_at(sc)
cases = [dict(g=g, cam=sc["cameras"][i], W=W, H=H, P=P, bg=sc["bg"], sh_degree=3) for i in range(V)]
dpix = np.random.default_rng(11).normal(size=(4, 3, H, W)).astype(np.float32)[:V]
settings = [util.gpu_settings(c) for c in cases]
names = ("means3D", "opacities", "shs", "scales", "rotations")


def run(flags):
    native.lib.csplat_debug_flags(flags)
    inp = util.gpu_inputs(cases[0])
    m2d = [torch.zeros(P, 3, device="cuda", requires_grad=True) for _ in range(V)]
    outs = []
    for i in range(V):      # one view at a time so that every view's own gradients can be read
        for k in names:
            inp[k].grad = None
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])]
        colors, _ = rasterize_views([settings[i]], kws, stacked=True)
        (colors * torch.tensor(dpix[i:i + 1], device="cuda")).sum().backward()
        torch.cuda.synchronize()
        outs.append(dict(mean2D=m2d[i].grad.cpu().numpy(), mean3D=inp["means3D"].grad.cpu().numpy(), opacity=inp["opacities"].grad.reshape(-1).cpu().numpy(),
                         sh=inp["shs"].grad.cpu().numpy(), scale=inp["scales"].grad.cpu().numpy(), rot=inp["rotations"].grad.cpu().numpy()))
    native.lib.csplat_debug_flags(0)
    return outs


hip = run(0)
hip_det = run(256)
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))  # noqa: E731
for i, case in enumerate(cases):
    o32, o64 = oracle_forward(case), oracle_forward(case, dtype=np.float64)
    g32, g64 = util.ro.backward(o32, dpix[i]), util.ro.backward(o64, dpix[i])
    print(f"view {i}: R={o64.R}")
    for k in ("mean2D", "mean3D", "opacity", "sh", "scale", "rot"):
        ref = np.asarray(getattr(g64, k), np.float64)
        e_h, e_d, e_o = rel(hip[i][k], ref), rel(hip_det[i][k], ref), rel(getattr(g32, k), ref)
        d = np.abs(np.asarray(hip[i][k], np.float64) - ref).reshape(P, -1).max(1)
        j = int(d.argmax())
        scl = float(np.abs(ref).max())
        # how many Gaussians are off by more than 1e-4 of the scale, and the same for the fp32 oracle
        n_h = int((d > 1e-4 * scl).sum())
        d32 = np.abs(np.asarray(getattr(g32, k), np.float64) - ref).reshape(P, -1).max(1)
        n_o = int((d32 > 1e-4 * scl).sum())
        print(f"  {k:8s} hip {e_h:.2e} hip_det {e_d:.2e} oracle32 {e_o:.2e} | >1e-4: hip {n_h} oracle32 {n_o} | worst id {j}: hip {np.ravel(hip[i][k].reshape(P, -1)[j])[:3]} "
              f"o64 {np.ravel(ref.reshape(P, -1)[j])[:3]} o32 {np.ravel(np.asarray(getattr(g32, k)).reshape(P, -1)[j])[:3]}")
        if k == "mean2D":
            s = g["scales"][j]
            print(f"     worst Gaussian {j}: scales {s} (aniso {s.max() / s.min():.1f}) opacity {g['opacities'][j]} radius {o64.radii[j]} depth {o64.depth[j]:.3f} "
                  f"conic {o64.conic_opacity[j]} tiles {o64.tiles_touched[j]}")
