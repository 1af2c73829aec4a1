"""Drop-in for `diff_gaussian_rasterization` (the depth fork), MI355X-native.

Mirrors the interface the reference binds at gaussian_renderer/__init__.py:16 and calls at :61-76,156-164:
    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                                  projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None) -> (color, radii, depth)
Gradients are delivered for means3D, means2D (NDC-space screen gradient used by densification,
scene_reconstruction/train_utils.py:290-292), shs, colors_precomp, opacities, scales, rotations, cov3D_precomp.
The depth image carries no gradient (as upstream).  All compute is in libcsplat.so (csplat_forward / csplat_backward).
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from csplat import native as _n


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t, device):
    if t is None:
        return None
    if t.device != device or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t.contiguous()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        _n.require_cuda(means3D)
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        means3D = _f32c(means3D, dev); opacities = _f32c(opacities, dev)
        sh = _f32c(sh, dev)
        colors_precomp = _f32c(colors_precomp, dev)
        scales = _f32c(scales, dev)
        rotations = _f32c(rotations, dev)
        cov3Ds_precomp = _f32c(cov3Ds_precomp, dev)
        bg = _f32c(rs.bg, dev); view = _f32c(rs.viewmatrix, dev); proj = _f32c(rs.projmatrix, dev)
        campos = _f32c(rs.campos, dev)
        M = int(sh.shape[1]) if sh is not None else 0
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        alloc = _n.ChunkAllocator(dev)
        R = C.c_int(0)
        geom, binning, image = C.c_void_p(), C.c_void_p(), C.c_void_p()
        with torch.cuda.device(dev):
            rc = _n.lib.csplat_forward(
                _n.stream_handle(dev), P, int(rs.sh_degree), M, _n.ptr(bg), W, H, _n.ptr(means3D), _n.ptr(sh),
                _n.ptr(colors_precomp), _n.ptr(opacities), _n.ptr(scales), float(rs.scale_modifier), _n.ptr(rotations),
                _n.ptr(cov3Ds_precomp), _n.ptr(view), _n.ptr(proj), _n.ptr(campos), float(rs.tanfovx), float(rs.tanfovy),
                int(bool(rs.prefiltered)), alloc.cb, None, _n.ptr(color), _n.ptr(depth), _n.ptr(radii), C.byref(R),
                C.byref(geom), C.byref(binning), C.byref(image))
        _n.check(rc, "csplat_forward")
        ctx.rs = rs
        ctx.num_rendered = int(R.value)
        ctx.M = M
        ctx.chunks = (alloc.chunks[_n_GEOM], alloc.chunks[_n_BINNING], alloc.chunks[_n_IMAGE])
        ctx.consts = (bg, view, proj, campos)
        ctx.save_for_backward(means3D, sh, colors_precomp, scales, rotations, cov3Ds_precomp, radii, color)
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth):
        means3D, sh, colors_precomp, scales, rotations, cov3Ds_precomp, radii, color = ctx.saved_tensors
        rs = ctx.rs
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        M = ctx.M
        bg, view, proj, campos = ctx.consts
        geom, binning, image = ctx.chunks
        grad_color = _f32c(grad_color, dev)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        d_mean2D, d_conic, d_opac, d_color = new(P, 3), new(P, 4), new(P, 1), new(P, 3)
        d_mean3D, d_cov3D = new(P, 3), new(P, 6)
        d_sh = new(P, M, 3) if sh is not None else None
        d_scale = new(P, 3) if scales is not None else None
        d_rot = new(P, 4) if rotations is not None else None
        scratch = torch.empty(max(int(_n.lib.csplat_backward_scratch_bytes(P, ctx.num_rendered)), 256), dtype=torch.uint8,
                              device=dev)
        with torch.cuda.device(dev):
            rc = _n.lib.csplat_backward(
                _n.stream_handle(dev), P, int(rs.sh_degree), M, ctx.num_rendered, _n.ptr(bg), W, H, _n.ptr(means3D),
                _n.ptr(sh), _n.ptr(colors_precomp), _n.ptr(scales), float(rs.scale_modifier), _n.ptr(rotations),
                _n.ptr(cov3Ds_precomp), _n.ptr(view), _n.ptr(proj), _n.ptr(campos), float(rs.tanfovx), float(rs.tanfovy),
                _n.ptr(radii), _n.ptr(geom), _n.ptr(binning), _n.ptr(image), _n.ptr(color), _n.ptr(grad_color),
                _n.ptr(scratch),
                _n.ptr(d_mean2D),
                _n.ptr(d_conic), _n.ptr(d_opac), _n.ptr(d_color), _n.ptr(d_mean3D), _n.ptr(d_cov3D), _n.ptr(d_sh),
                _n.ptr(d_scale), _n.ptr(d_rot))
        _n.check(rc, "csplat_backward")
        return (d_mean3D, d_mean2D, d_sh, d_color if colors_precomp is not None else None, d_opac, d_scale, d_rot,
                d_cov3D if cov3Ds_precomp is not None else None, None)


_n_GEOM, _n_BINNING, _n_IMAGE = 0, 1, 2


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of the upstream extension: visible iff view-space z > 0.2."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix.to(positions.device, torch.float32)
            z = positions @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
