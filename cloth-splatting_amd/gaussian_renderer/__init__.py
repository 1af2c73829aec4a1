"""Drop-in for /root/reference/gaussian_renderer/__init__.py: render() with the reference's signature and
RenderResults record (:22-36, :39-206), driving the MI355X rasterizer (libcsplat.so) through the
diff_gaussian_rasterization drop-in.  `pc` may be the reference's own MultiGaussianMesh or csplat.gaussians.MeshGaussians;
`simulator` any module with forward(time_vector=[V,1]) -> [V,3] (ResidualMeshSimulator).
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views


class RenderResults(NamedTuple):
    render: torch.Tensor
    viewspace_points: torch.Tensor
    visibility_filter: torch.Tensor
    radii: torch.Tensor
    depth: torch.Tensor
    means3D_deform: torch.Tensor
    vertice_deform: torch.Tensor
    shadows_mean: Optional[torch.Tensor]
    shadows_std: Optional[torch.Tensor]
    projections: torch.Tensor
    rotations: torch.Tensor
    opacities: torch.Tensor
    shadows: Optional[torch.Tensor]
    vertice_projections: Optional[torch.Tensor]


_SIZES = {}
_ZEROS = {}


def _screenspace_zeros(n_views, like):
    """[n_views, P, 3] zeros for the screen-space leaves of a step.  Nothing ever writes into them (the rasterizer reads no value of
    means2D, it only returns its gradient; every camera's leaf is a detached view), so ONE buffer per (device, dtype) serves every step
    instead of a fill launch per step; re-made when the step needs more room (densification)."""
    key = (like.device, like.dtype)
    need = n_views * int(like.shape[0]) * 3
    buf = _ZEROS.get(key)
    if buf is None or buf.numel() < need:
        buf = _ZEROS[key] = torch.zeros(need + need // 4, dtype=like.dtype, device=like.device)
    return buf[:need].view(n_views, int(like.shape[0]), 3)


def _project_torch(full, W, H, points):
    # columns of [p, 1] @ full as three broadcast multiply-adds: the [P,4] x [4,4] product is a skinny GEMM that the BLAS
    # runs in 72 us at P = 100k (more than K1 + K8 of the same view)
    hom = torch.addcmul(torch.addcmul(torch.addcmul(full[3], points[:, 0:1], full[0]), points[:, 1:2], full[1]),
                        points[:, 2:3], full[2])
    ndc = hom[:, :2] / hom[:, 3:4]
    if W == H:
        return ((ndc + 1.0) * float(W) - 1.0) * 0.5
    size = _SIZES.get((W, H, points.device))
    if size is None:   # (uploaded once per image size: a host->device copy stalls the stream)
        size = _SIZES[(W, H, points.device)] = torch.tensor([float(W), float(H)], device=points.device)
    return ((ndc + 1.0) * size - 1.0) * 0.5


class _ProjectPoints(torch.autograd.Function):
    """csplat_project_points: the projections by-product in one launch.  Its gradient is rarely asked for (nothing in the
    reference's losses uses the projections): backward re-evaluates the torch formulation under autograd."""

    @staticmethod
    def forward(ctx, points, full, W, H):
        from csplat import native as _n
        pts = points.contiguous().float()
        out = torch.empty(pts.shape[0], 2, dtype=torch.float32, device=pts.device)
        with _n.on_device(pts.device):
            _n.check(_n.lib.csplat_project_points(_n.stream_handle(pts.device), pts.shape[0], _n.ptr(full), W, H, _n.ptr(pts),
                                                  _n.ptr(out)), "csplat_project_points")
        ctx.save_for_backward(points, full)
        ctx.size = (W, H)
        return out

    @staticmethod
    def backward(ctx, g):
        points, full = ctx.saved_tensors
        with torch.enable_grad():
            p = points.detach().requires_grad_()
            out = _project_torch(full, ctx.size[0], ctx.size[1], p)
        return torch.autograd.grad(out, p, g)[0], None, None, None


def _project(cam, points):
    """pixel coordinates of world points (reference :166-179): p_h @ full_proj, /w, ndc -> ((v+1)*S-1)/2."""
    full = cam.full_proj_transform.to(points.device)
    W, H = int(cam.image_width), int(cam.image_height)
    if points.is_cuda and points.dtype == torch.float32 and full.dtype == torch.float32 and full.is_contiguous():
        return _ProjectPoints.apply(points, full, W, H)
    return _project_torch(full, W, H, points)


def _prepare(viewpoint_camera, pc, simulator, pipe, bg_color, scaling_modifier, override_color, log_deform_path,
             render_static, shared=None, vertice_deform=None, transformed=None):
    """everything of render() up to the rasterizer call: settings, rasterizer keyword arguments, by-products.
    `shared` carries the view-independent activations (features, opacity, scaling) so that several views of one step
    pass the SAME tensor objects to the rasterizer (one gradient buffer for all of them, see rasterize_views);
    `vertice_deform` is the simulator's output for this camera when the caller evaluated all cameras' times at once."""
    if shared is None:
        shared = {}
    if "opacity" not in shared:
        shared["opacity"] = pc.get_opacity
    # (the reference builds the zero tensor below from pc.get_xyz -- a gather + barycentric blend of the REST mesh per call;
    # only its shape is needed unless the static pose itself is rendered or logged)
    base_xyz = pc.get_xyz() if (render_static or log_deform_path is not None) else None
    dev = shared["opacity"].device
    # zero tensor whose gradient is the screen-space (NDC) gradient of the 2D means (used by densification)
    # (upstream writes zeros_like(xyz, requires_grad=True) + 0 and retain_grad(): a non-leaf whose .grad is kept by a Python
    # hook.  A leaf gives the same .grad to train_step / densification without the extra launch and hook per camera.)
    if "screenspace_pool" in shared:   # render_views: ONE zero fill for all cameras, every camera's leaf is a slice of it
        screenspace_points = shared["screenspace_pool"].pop().detach().requires_grad_()
    else:
        screenspace_points = torch.zeros(shared["opacity"].shape[0], 3, dtype=shared["opacity"].dtype, requires_grad=True, device=dev)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform.to(dev), projmatrix=viewpoint_camera.full_proj_transform.to(dev),
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center.to(dev), prefiltered=False, debug=False)

    if "features" not in shared:
        if pipe.compute_cov3D_python:
            shared["cov3D"], shared["scales"] = pc.get_covariance(scaling_modifier), None
        else:
            shared["cov3D"], shared["scales"] = None, pc.get_scaling
        shared["features"] = pc.get_features if override_color is None else None
    cov3D_precomp, scales, opacity = shared["cov3D"], shared["scales"], shared["opacity"]

    if render_static:
        vertice_deform = pc.mesh.pos
        means3D_deform = base_xyz
        rotations_deform = pc.get_rotation()
    else:
        if vertice_deform is None:
            time = torch.tensor(viewpoint_camera.time).to(pc.mesh.pos.device).repeat(pc.mesh.pos.shape[0], 1)
            vertice_deform = simulator(time_vector=time)
        if transformed is not None:      # (xyz, rotation) of this camera from the all-cameras transform of render_views
            means3D_deform, rotations_deform = transformed
        else:
            means3D_deform = pc.get_xyz(vertice_deform)
            rotations_deform = pc.get_rotation(vertice_deform)

    if log_deform_path is not None:
        np.savez(log_deform_path, means3D=base_xyz.detach().cpu().numpy(),
                 means3D_deform=means3D_deform.detach().cpu().numpy(),
                 vertice_deform=vertice_deform.detach().cpu().numpy(), rotations=rotations_deform.detach().cpu().numpy(),
                 vertice_rotations=pc.get_vertice_rotation(vertice_deform).detach().cpu().numpy())

    # shadow scalars are disabled in the reference (always None): colours come from SH inside the rasterizer
    shs, colors_precomp = (shared["features"], None) if override_color is None else (None, override_color)
    # (as the reference, :156-164: `rotations` is passed in the python-covariance branch too -- upstream's rasterizer then raises
    # its "exactly one of either scale/rotation pair or precomputed 3D covariance" error, and so does the drop-in;
    # tests/golden/render_wiring.npz holds what the reference's own render() hands over in every branch)
    kwargs = dict(means3D=means3D_deform, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                  opacities=opacity, scales=scales, rotations=rotations_deform, cov3D_precomp=cov3D_precomp)
    return raster_settings, kwargs, (screenspace_points, means3D_deform, vertice_deform, rotations_deform, opacity)


def _package(viewpoint_camera, raster_out, extras, project_vertices, by_products=True):
    rendered_image, radii, depth = raster_out
    screenspace_points, means3D_deform, vertice_deform, rotations_deform, opacity = extras
    # by_products=False (render_views for a caller that only trains on the record, csplat.train.train_step): the projections and
    # the per-view visibility mask -- two launches per view that nothing in the step reads -- are left out (None)
    gaussian_projections = _project(viewpoint_camera, means3D_deform) if by_products else None
    vertice_projections = _project(viewpoint_camera, vertice_deform) if project_vertices else None
    return RenderResults(render=rendered_image, viewspace_points=screenspace_points, visibility_filter=(radii > 0) if by_products else None,
                         radii=radii, depth=depth, means3D_deform=means3D_deform, vertice_deform=vertice_deform,
                         shadows_mean=None, shadows_std=None, projections=gaussian_projections,
                         rotations=rotations_deform, opacities=opacity, shadows=None,
                         vertice_projections=vertice_projections)


def render(viewpoint_camera, pc, simulator, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           log_deform_path=None, no_shadow=False, render_static=False, project_vertices=False) -> RenderResults:
    """Render the scene (background tensor must be on the GPU)."""
    settings, kwargs, extras = _prepare(viewpoint_camera, pc, simulator, pipe, bg_color, scaling_modifier, override_color,
                                        log_deform_path, render_static)
    out = GaussianRasterizer(raster_settings=settings)(**kwargs)
    return _package(viewpoint_camera, out, extras, project_vertices)


def render_views(viewpoint_cameras, pc, simulator, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
                 no_shadow=False, render_static=False, project_vertices=False, return_stacked=False, vertice_deforms=None,
                 by_products=True):
    """render() for every camera of a training step in one rasterizer call (diff_gaussian_rasterization.rasterize_views:
    one HIP stream per view, the views' kernels overlap, shared parameters get one gradient buffer).  Same results as
    [render(c, ...) for c in viewpoint_cameras]; no counterpart upstream, whose train loop renders camera by camera
    (scene_reconstruction/train_utils.py:204-260).  return_stacked=True also returns the [V,3,H,W] image batch (None when
    the cameras differ in size) so that the caller's losses need no torch.cat."""
    shared, prepared = {}, []
    viewpoint_cameras = list(viewpoint_cameras)
    deforms = vertice_deforms      # [T, V, 3] when the caller already evaluated the simulator for these cameras
    if deforms is None and not render_static and viewpoint_cameras and hasattr(simulator, "forward_times"):
        deforms = simulator.forward_times([cam.time for cam in viewpoint_cameras])   # [T, V, 3]: one pass over the output layer
    moved = acts = None
    want_acts = bool(viewpoint_cameras) and hasattr(pc, "activations") and override_color is None and not pipe.compute_cov3D_python
    if deforms is not None and deforms.is_cuda and getattr(pc, "fused", False) and hasattr(pc, "transform_views"):
        both = pc.step_inputs(deforms) if (want_acts and hasattr(pc, "step_inputs")) else None
        if both is not None:        # the transform of all cameras AND the activations: one autograd node, two launches each way
            moved, acts = (both[0], both[1]), both[2:]
        else:
            moved = pc.transform_views(deforms)     # mesh -> Gaussian transform of all cameras in one launch each way
    deform_views = None if deforms is None else deforms.unbind(0)
    if viewpoint_cameras:
        if acts is None:
            acts = pc.activations() if want_acts else None
        if acts is not None:    # sigmoid / exp / cat of the Gaussian parameters in one launch each way
            shared["opacity"], shared["scales"], shared["features"], shared["cov3D"] = acts[0], acts[1], acts[2], None
        else:
            shared["opacity"] = pc.get_opacity
        shared["screenspace_pool"] = list(_screenspace_zeros(len(viewpoint_cameras), shared["opacity"]).unbind(0))
    for i, cam in enumerate(viewpoint_cameras):
        prepared.append(_prepare(cam, pc, simulator, pipe, bg_color, scaling_modifier, override_color, None, render_static,
                                 shared, None if deforms is None else deform_views[i],
                                 None if moved is None else (moved[0][i], moved[1][i])))
    if not prepared:
        return ([], None) if return_stacked else []
    sizes = {(p[0].image_height, p[0].image_width) for p in prepared}
    if len(sizes) == 1:   # the images of the step in ONE [V,3,H,W] tensor: each RenderResults.render is a slice of it
        stacked, outs = rasterize_views([p[0] for p in prepared], [p[1] for p in prepared], stacked=True)
    else:
        stacked, outs = None, rasterize_views([p[0] for p in prepared], [p[1] for p in prepared])
    res = [_package(cam, out, p[2], project_vertices, by_products) for cam, out, p in zip(viewpoint_cameras, outs, prepared)]
    return (res, stacked) if return_stacked else res
