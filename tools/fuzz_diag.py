#!/usr/bin/env python3
"""fuzz_diag.py: for the wild fuzz scenes of tests/test_raster_gpu.py, compare GPU vs fp64 oracle AND fp32 oracle vs fp64
oracle -- separates conditioning (both fp32 paths drift alike) from defects (only the GPU drifts)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
import numpy as np
import util
src = open(os.path.join(ROOT, "tests", "test_raster_gpu.py")).read()
ns = {}
exec("import numpy as np\n" + src[src.index("def _wild_case(seed):"):src.index('@pytest.mark.parametrize("seed"')], ns)
import torch
import diff_gaussian_rasterization as dgr
for seed in [int(a) for a in sys.argv[1:]] or range(100, 112):
    case = ns["_wild_case"](seed)
    o32 = util.oracle_forward(case); o64 = util.oracle_forward(case, dtype=np.float64)
    dpix = np.random.default_rng(seed).normal(size=(3, case["H"], case["W"])).astype(np.float32)
    g64, g32 = util.ro.backward(o64, dpix), util.ro.backward(o32, dpix)
    inp = util.gpu_inputs(case); rs = util.gpu_settings(case)
    color, radii, depth = dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"],
                                                     shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
    (color * torch.tensor(dpix, device="cuda")).sum().backward()
    got = dict(mean3D=inp["means3D"].grad, mean2D=inp["means2D"].grad, opacity=inp["opacities"].grad.reshape(-1),
               sh=inp["shs"].grad, scale=inp["scales"].grad, rot=inp["rotations"].grad)
    print(f"seed {seed}: P={case['P']} {case['W']}x{case['H']} R={o32.R}  n_contrib mismatches gpu/o32 vs o64: "
          f"img err gpu {util.rel_err(color.detach().cpu().numpy(), o64.color):.2e} o32 {util.rel_err(o32.color, o64.color):.2e}")
    for k, v in got.items():
        a = v.cpu().numpy().astype(np.float64).reshape(case["P"], -1)
        b = np.asarray(getattr(g64, k), np.float64).reshape(case["P"], -1)
        c = np.asarray(getattr(g32, k), np.float64).reshape(case["P"], -1)
        s = np.abs(b).max() + 1e-30
        eg, eo = np.abs(a - b).max(1) / s, np.abs(c - b).max(1) / s
        worst = int(np.argmax(eg))
        print(f"   {k:8s} gpu: max {eg.max():.2e} n>1e-4 {int((eg > 1e-4).sum()):4d} | fp32 oracle: max {eo.max():.2e} n>1e-4 "
              f"{int((eo > 1e-4).sum()):4d} | both bad {int(((eg > 1e-4) & (eo > 1e-4)).sum()):4d} | worst id {worst} "
              f"scales {case['g']['scales'][worst]} radius {int(o32.radii[worst])}")
        odd = np.nonzero((eg > 1e-4) & (eo <= 2.5e-5))[0]
        for j in odd[:4]:
            print(f"      gpu-only: id {j} eg {eg[j]:.2e} eo {eo[j]:.2e} scales {case['g']['scales'][j]} radius {int(o32.radii[j])} "
                  f"opacity {float(case['g']['opacities'][j]):.4f} gpu {a[j][:3]} o64 {b[j][:3]} o32 {c[j][:3]}")
