"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cloth-splatting_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from csplat import synthetic as syn  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def make_case(P=2000, W=128, H=96, seed=7, grid=20, scale_mul=1.0, theta=0.0, radius=4.0, fovx=syn.CAMERA_ANGLE_X):
    sc = syn.scene_1(P=P, W=W, H=H, n_cams=1, grid=grid, seed=seed)
    g = syn.gaussians_at(sc)
    g["scales"] = (g["scales"] * scale_mul).astype(np.float32)
    cam = syn.make_camera(theta, W, H, radius=radius, fovx=fovx)
    return dict(g=g, cam=cam, W=W, H=H, P=P, bg=sc["bg"], sh_degree=3)


def oracle_forward(case, dtype=np.float32, **over):
    g, cam = case["g"], case["cam"]
    kw = dict(shs=g["shs"], sh_degree=case["sh_degree"], scales=g["scales"], rotations=g["rotations"])
    kw.update(over)
    return ro.forward(g["means3D"], g["opacities"], cam["world_view_transform"], cam["full_proj_transform"],
                      cam["camera_center"], cam["tanfovx"], cam["tanfovy"], case["W"], case["H"], case["bg"],
                      dtype=dtype, **kw)


def closed_form_weights(module, salt=0):
    """the weights of tests/golden/make_golden.py:closed_form_weights (identical arithmetic: an integer hash -> float64 -> float32),
    so that gnn128.npz pins a 128-wide network without storing its weights"""
    import torch
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            i = np.arange(p.numel(), dtype=np.uint64)
            h = ((i * np.uint64(2654435761) + np.uint64(40503 * (k + 1) + 977 * salt)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32) - 0.5
            if p.dim() == 2:
                v = h * (2.0 / np.sqrt(p.shape[1]))
            elif name.endswith("weight"):
                v = 1.0 + 0.2 * h
            else:
                v = 0.2 * h
            p.copy_(torch.from_numpy(v.astype(np.float32)).reshape(p.shape).to(p.device))
    return module


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rowwise_rel_err(a, b, rows, floor=1e-2):
    """ELEMENT-WISE relative error with a floor, per Gaussian (`rel_err` above is max-abs error over the tensor's
    max -- an error of 1e-4 of the LARGEST gradient could be 100 % of a small one).  For row i (one Gaussian's gradient entries):
        e_i = max_j |a_ij - b_ij| / max(max_j |b_ij|, floor * max |b|)
    i.e. relative to the Gaussian's own gradient magnitude, except that Gaussians whose gradient is below `floor` of the tensor's
    largest are measured against that floor (a gradient that is itself a rounding-level sum has no relative accuracy to speak of).
    Returns the vector e [rows]."""
    a = np.asarray(a, np.float64).reshape(rows, -1); b = np.asarray(b, np.float64).reshape(rows, -1)
    scale = np.abs(b).max() + 1e-30
    den = np.maximum(np.abs(b).max(1), floor * scale)
    return np.abs(a - b).max(1) / den


def image_err(a, b, outlier_frac=1e-4, outlier_tol=1e-2):
    """Relative max error of an image, robust to THRESHOLD TIES: the compositing rule is discontinuous (a Gaussian is
    dropped at a pixel when alpha < 1/255, a pixel stops when T(1-alpha) < 1e-4), so fp32-vs-fp64 (or v_exp_f32-vs-expf)
    can legitimately flip one such decision at an isolated pixel, changing it by up to ~alpha*colour*T <= 1/255-ish.
    At most `outlier_frac` of the pixels may be such ties, each bounded by `outlier_tol`; the returned value is the
    relative error over all other pixels."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    d = np.abs(a - b) / scale
    per_px = d.reshape(-1, d.shape[-2] * d.shape[-1]).max(0) if d.ndim == 3 else d.reshape(-1)
    k = int(np.ceil(outlier_frac * per_px.size))
    srt = np.sort(per_px)
    assert srt[-1] <= outlier_tol, f"pixel error {srt[-1]:.3e} exceeds the threshold-tie bound {outlier_tol}"
    return float(srt[-(k + 1)])


# ---------------------------------------------------------------- GPU side (imports torch lazily)
def gpu_settings(case, scale_mod=1.0, sh_degree=None, device="cuda"):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = case["cam"]
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)  # noqa: E731
    return GaussianRasterizationSettings(
        image_height=case["H"], image_width=case["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=t(case["bg"]), scale_modifier=scale_mod, viewmatrix=t(cam["world_view_transform"]),
        projmatrix=t(cam["full_proj_transform"]), sh_degree=case["sh_degree"] if sh_degree is None else sh_degree,
        campos=t(cam["camera_center"]), prefiltered=False, debug=False)


def gpu_inputs(case, device="cuda", requires_grad=True):
    import torch
    g = case["g"]
    mk = lambda a: torch.tensor(np.asarray(a, np.float32), device=device, requires_grad=requires_grad)  # noqa: E731
    d = dict(means3D=mk(g["means3D"]), opacities=mk(g["opacities"]), shs=mk(g["shs"]), scales=mk(g["scales"]),
             rotations=mk(g["rotations"]))
    d["means2D"] = torch.zeros(case["P"], 3, device=device, requires_grad=requires_grad)
    return d


def gpu_chunks(ctx_chunks, P, W, H, R):
    """Decode the GEOM / BINNING / IMAGE byte chunks (layout from the C-ABI) into numpy arrays."""
    from csplat import native as n
    geom, binning, image = (c.cpu().numpy() for c in ctx_chunks)
    o8 = (C.c_size_t * 8)(); n.lib.csplat_geom_layout(P, o8)
    o2 = (C.c_size_t * 2)(); n.lib.csplat_binning_layout(R, W, H, o2)
    o3 = (C.c_size_t * 3)(); n.lib.csplat_image_layout(W, H, o3)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def view(buf, off, dt, count):
        return np.frombuffer(buf.tobytes()[off:off + count * np.dtype(dt).itemsize], dtype=dt).copy()
    out = dict(
        depth=view(geom, o8[0], np.float32, P), xy=view(geom, o8[1], np.float32, 2 * P).reshape(P, 2),
        conic_opacity=view(geom, o8[2], np.float32, 4 * P).reshape(P, 4), rgb=view(geom, o8[3], np.float32, 3 * P).reshape(P, 3),
        cov3D=view(geom, o8[4], np.float32, 6 * P).reshape(P, 6), clamped=view(geom, o8[5], np.uint32, P),
        tiles_touched=view(geom, o8[6], np.uint32, P), offsets=view(geom, o8[7], np.uint32, P),
        keys=view(binning, o2[0], np.uint64, R), ids=view(binning, o2[1], np.uint32, R),
        ranges=view(image, o3[0], np.int32, 2 * tiles).reshape(tiles, 2),
        n_contrib=view(image, o3[1], np.uint32, W * H).reshape(H, W), final_T=view(image, o3[2], np.float32, W * H).reshape(H, W))
    out["_binning_raw"] = ctx_chunks[1]          # (tools/k6_stats.py decodes the masks / "blended" words from it)
    return out


def gpu_forward_raw(case, inputs=None, settings=None, **over):
    """Run the autograd Function directly so the saved chunks can be inspected.  Returns (color, radii, depth, state)."""
    import torch
    import diff_gaussian_rasterization as dgr
    inp = inputs or gpu_inputs(case)
    rs = settings or gpu_settings(case)
    args = dict(sh=inp["shs"], colors_precomp=None, scales=inp["scales"], rotations=inp["rotations"], cov3Ds_precomp=None)
    args.update(over)

    class Ctx:  # minimal stand-in for the autograd ctx so that forward() can be called un-differentiated
        def save_for_backward(self, *a): self.saved_tensors = a
        def mark_non_differentiable(self, *a): pass
    ctx = Ctx()
    with torch.no_grad():
        color, radii, depth = dgr._RasterizeGaussians.forward(
            ctx, inp["means3D"], inp["means2D"], args["sh"], args["colors_precomp"], inp["opacities"], args["scales"],
            args["rotations"], args["cov3Ds_precomp"], rs)
    torch.cuda.synchronize()
    vs = ctx.view_state
    state = gpu_chunks(vs.chunks, case["P"], case["W"], case["H"], vs.num_rendered)
    state["R"] = vs.num_rendered
    return color, radii, depth, state


# ---------------------------------------------------------------- CPU stand-in for the rasterizer (oracle-backed, tests only)
def oracle_raster_function():
    """torch.autograd.Function over oracle/raster_ref.c (fp64 build): forward AND analytic backward, incl. the NDC-space
    gradient of means2D.  Lets the host logic around the rasterizer (train_step, view-parallel sharding) run on CPU tensors in
    the world_size-2 gloo tests; the product has no CPU path."""
    import torch

    class OracleRaster(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, opacity, shs, scales, rots, cam, bg_np, sh_degree):
            n = lambda t: t.detach().numpy()  # noqa: E731
            H, W = int(cam.image_height), int(cam.image_width)
            o = ro.forward(n(means3D), n(opacity), n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center),
                           np.tan(cam.FoVx * 0.5), np.tan(cam.FoVy * 0.5), W, H, bg_np, shs=n(shs), sh_degree=sh_degree,
                           scales=n(scales), rotations=n(rots), dtype=np.float64)
            ctx.o = o
            radii = torch.from_numpy(o.radii.copy())
            ctx.mark_non_differentiable(radii)
            return torch.from_numpy(o.color.copy()), radii

        @staticmethod
        def backward(ctx, g_color, _g_radii):
            g = ro.backward(ctx.o, g_color.contiguous().numpy())
            t = torch.from_numpy
            return t(g.mean3D), t(g.mean2D), t(g.opacity).reshape(-1, 1), t(g.sh), t(g.scale), t(g.rot), None, None, None
    return OracleRaster


def oracle_render_views(cams, pc, sim, pipe, bg, render_static=False, return_stacked=True, vertice_deforms=None, by_products=True):
    """gaussian_renderer.render_views on CPU fp64 tensors through the oracle (same return convention)."""
    import torch
    from types import SimpleNamespace
    F = oracle_raster_function()
    bg_np = np.asarray(bg.detach().numpy(), np.float64)
    res = []
    for i, cam in enumerate(cams):
        if render_static:
            verts = pc.mesh.pos
        elif vertice_deforms is not None:
            verts = vertice_deforms[i]
        else:
            verts = sim(time_vector=torch.tensor(cam.time, dtype=pc.mesh.pos.dtype).repeat(pc.mesh.pos.shape[0], 1))
        xyz = pc.get_xyz(verts if not render_static else None)
        rots = pc.get_rotation(verts if not render_static else None)
        m2d = torch.zeros(xyz.shape[0], 3, dtype=xyz.dtype, requires_grad=True)
        color, radii = F.apply(xyz, m2d, pc.get_opacity, pc.get_features, pc.get_scaling, rots, cam, bg_np, pc.active_sh_degree)
        res.append(SimpleNamespace(render=color, radii=radii, visibility_filter=radii > 0, viewspace_points=m2d, vertice_deform=verts))
    return (res, None) if return_stacked else res
