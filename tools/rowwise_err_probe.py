#!/usr/bin/env python3
"""Calibration of the per-Gaussian relative gradient check (tests/util.py:rowwise_rel_err): quantiles of the row-wise error of the HIP
gradients and of the fp32 ORACLE's gradients against the fp64 oracle, for the small test cases and several floors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
import test_raster_gpu as tr  # noqa: E402

for cfg in tr.CASES:
    case = util.make_case(**cfg)
    P = case["P"]
    dpix = np.random.default_rng(3).normal(size=(3, case["H"], case["W"])).astype(np.float32)
    o64 = util.oracle_forward(case, dtype=np.float64); g64 = util.ro.backward(o64, dpix)
    o32 = util.oracle_forward(case); g32 = util.ro.backward(o32, dpix)
    inp, kw, color, radii, depth = tr._run_gpu(case, dpix)
    got = dict(mean3D=inp["means3D"].grad, mean2D=inp["means2D"].grad, opacity=inp["opacities"].grad.reshape(-1),
               sh=inp["shs"].grad, scale=inp["scales"].grad, rot=inp["rotations"].grad)
    for k, v in got.items():
        for floor in (1e-1, 1e-2, 1e-3):
            e = util.rowwise_rel_err(v.cpu().numpy(), getattr(g64, k), P, floor)
            e32 = util.rowwise_rel_err(getattr(g32, k), getattr(g64, k), P, floor)
            q = lambda x: " ".join(f"{np.quantile(x, t):.1e}" for t in (0.5, 0.99, 0.999, 1.0))  # noqa: E731
            print(f"P={P} {k:8s} floor={floor:.0e} hip: {q(e)}  n>1e-4: {(e > 1e-4).sum():4d} | oracle32: {q(e32)} n>1e-4: {(e32 > 1e-4).sum():4d}", flush=True)
