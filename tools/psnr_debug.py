"""tools/psnr_debug.py [step] -- follow the HIP training of tests/test_psnr_parity_gpu.py to `step`, then take the rasterizer inputs of
each camera at that state and compare the raw rasterizer backward (HIP, in several modes) with the fp32 and fp64 oracle on the SAME
inputs and the same dL/dimage; the inputs are saved to gpurun_out/psnr_debug_<step>.npz.  GPU box."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import test_psnr_parity_gpu as t
import util
from csplat import native, train as tr
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from oracle import raster_oracle as ro

S = int(sys.argv[1]) if len(sys.argv) > 1 else 425
saved = {}


def hook(it, pc, sim, cams, bg, build, psnr):
    if it != S:
        return
    with torch.no_grad():
        for k, c in enumerate(cams):
            V = pc.mesh.pos.shape[0]
            verts = sim(time_vector=torch.tensor(c.time, device="cuda").repeat(V, 1))
            saved[k] = dict(means3D=pc.get_xyz(verts).cpu().numpy(), rotations=pc.get_rotation(verts).cpu().numpy(), opacities=pc.get_opacity.cpu().numpy(),
                            shs=pc.get_features.cpu().numpy(), scales=pc.get_scaling.cpu().numpy(), view=c.world_view_transform.cpu().numpy(),
                            proj=c.full_proj_transform.cpu().numpy(), campos=c.camera_center.cpu().numpy(), fovx=c.FoVx, fovy=c.FoVy,
                            W=c.image_width, H=c.image_height, gt=c.original_image.cpu().numpy())


t.STEP_HOOK[0] = hook
t.run_parity(False, S, hip_only=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/psnr_debug_{S}.npz", **{f"{k}.{n}": np.asarray(v) for k, d in saved.items() for n, v in d.items()})
rel = lambda a, b, s=None: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / ((np.abs(b).max() if s is None else s) + 1e-30))  # noqa: E731
for k, d in saved.items():
    W, H = int(d["W"]), int(d["H"])
    tfx, tfy = np.tan(d["fovx"] * 0.5), np.tan(d["fovy"] * 0.5)
    bgn = np.ones(3, np.float32)
    outs = {}
    for dt in (np.float32, np.float64):
        o = ro.forward(d["means3D"], d["opacities"], d["view"], d["proj"], d["campos"], tfx, tfy, W, H, bgn, shs=d["shs"], sh_degree=3, scales=d["scales"],
                       rotations=d["rotations"], dtype=dt)
        # dL/dimage of the step's image loss at THIS image (L1 + 0.05 D-SSIM against the camera's target), evaluated once from the fp64 image
        if dt == np.float32:
            img = torch.tensor(o.color.astype(np.float64), requires_grad=True)
            tr.image_losses(img[None], torch.tensor(d["gt"].astype(np.float64))[None], tr.DEFAULT_OPT).backward()
            dpix = img.grad.numpy().astype(np.float32)
        outs[dt] = (o, ro.backward(o, dpix))
    o32, g32 = outs[np.float32]
    o64, g64 = outs[np.float64]
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)  # noqa: E731
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=T(bgn), scale_modifier=1.0, viewmatrix=T(d["view"]),
                                       projmatrix=T(d["proj"]), sh_degree=3, campos=T(d["campos"]), prefiltered=False, debug=False)
    for flags in (256, 0, 8192, 256 | 8192):
        native.lib.csplat_debug_flags(flags)
        inp = {n: T(d[n], True) for n in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2d = torch.zeros(d["means3D"].shape[0], 3, device="cuda", requires_grad=True)
        color, radii, depth = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                                     rotations=inp["rotations"])
        (color * T(dpix)).sum().backward()
        torch.cuda.synchronize()
        native.lib.csplat_debug_flags(0)
        got = dict(mean3D=inp["means3D"].grad, mean2D=m2d.grad, opacity=inp["opacities"].grad.reshape(-1), sh=inp["shs"].grad, scale=inp["scales"].grad,
                   rot=inp["rotations"].grad)
        line = []
        for n, v in got.items():
            v = v.cpu().numpy()
            r32, r64 = np.asarray(getattr(g32, n)), np.asarray(getattr(g64, n))
            P = v.shape[0]
            dd = np.abs(v.reshape(P, -1).astype(np.float64) - r32.reshape(P, -1)).max(1)
            j = int(dd.argmax())
            line.append(f"{n}: hip-o32 {rel(v, r32, np.abs(r64).max()):.1e} o32-o64 {rel(r32, r64):.1e} worst {j}")
        print(f"cam {k} flags {flags:5d} image hip-o32 {rel(color.detach().cpu().numpy(), o32.color):.1e} | " + " | ".join(line), flush=True)
    st_mism = None
