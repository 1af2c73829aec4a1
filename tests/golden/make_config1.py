"""tests/golden/make_config1.py -- BASELINE.json configs[0] / BASELINE.md section 2 row 1: synthetic scene_1, P = 5,000
Gaussians, 1 camera 400x400, SH degree 3 -- the CPU RESTATEMENT (oracle/raster_ref.c, fp32 build) forward + backward, dumped
as the fixture tests/golden/config1.npz: tile ranges, sorted instance list, n_contrib, RGB, depth, final_T and every
gradient, next to the inputs they were computed from (so that a maintainer holding the upstream CUDA extension can run the
same Gaussians through it and compare).  This is NOT a reference-generated vector (the rasterizer's sources are absent from
the reference): it pins the oracle against drift and gives the HIP path a fixed config-1 target.

    python tests/golden/make_config1.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cloth-splatting_amd"), ROOT):
    sys.path.insert(0, p)
from csplat import synthetic as syn  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402

P, W, H = 5000, 400, 400


def config1():
    sc = syn.scene_1(P=P, W=W, H=H, n_cams=1)
    return sc, syn.gaussians_at(sc), sc["cameras"][0]


def dpix():
    return np.random.default_rng(6666).normal(size=(3, H, W)).astype(np.float32)


def run_oracle(dtype=np.float32):
    sc, g, cam = config1()
    o = ro.forward(g["means3D"], g["opacities"], cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                   cam["tanfovx"], cam["tanfovy"], W, H, sc["bg"], shs=g["shs"], sh_degree=3, scales=g["scales"],
                   rotations=g["rotations"], dtype=dtype)
    gr = ro.backward(o, dpix())
    return sc, g, cam, o, gr


if __name__ == "__main__":
    ro.build()
    t0 = time.perf_counter()
    sc, g, cam, o, gr = run_oracle()
    dt = time.perf_counter() - t0
    out = dict(means3D=g["means3D"], opacities=g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
               view=cam["world_view_transform"], proj=cam["full_proj_transform"], campos=cam["camera_center"],
               tanfov=np.array([cam["tanfovx"], cam["tanfovy"]]), bg=sc["bg"],
               R=np.array(o.R), radii=o.radii, tiles_touched=o.tiles_touched, keys=o.keys, ids=o.ids, ranges=o.ranges,
               n_contrib=o.n_contrib, color=o.color, depth=o.out_depth, final_T=o.final_T,
               d_mean2D=gr.mean2D, d_mean3D=gr.mean3D, d_opacity=gr.opacity, d_sh=gr.sh, d_scale=gr.scale, d_rot=gr.rot)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1.npz"), **out)
    print(f"config1: R = {o.R}, visible = {int((o.radii > 0).sum())}, oracle fwd+bwd {dt * 1e3:.1f} ms on {ro.num_threads()} threads "
          f"= {W * H / dt / 1e6:.2f} Mpix/s")
