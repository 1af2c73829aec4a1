"""MeshGaussians: the hot-path subset of the reference's MultiGaussianMesh (scene_reconstruction/gaussian_mesh.py)
that gaussian_renderer.render() and train_step touch -- parameters, activations (gaussian_model.py:27-48,96-121),
get_xyz (barycentric gather, gaussian_mesh.py:151-169), get_rotation (per-Gaussian Kabsch of its face + quaternion
composition, :171-188, incl. the wxyz/xyzw convention mix of SURVEY F8, reproduced not fixed) and the
distCUDA2-based scale initialisation (:249-251).  Densification / pruning / Adam-state surgery come from
csplat/densify.py (DensifyMixin), point_cloud.ply I/O from csplat/ply.py (N4)."""
from types import SimpleNamespace

import torch
from torch import nn

from . import native as _n
from . import rotations as rot
from .native import require_cuda


class MeshTransform(torch.autograd.Function):
    """(deformed vertices, face_bary, _rotation) -> (means3D, rotations): one HIP kernel each way
    (csplat_mesh_transform_fwd_views / _bwd_views, include/csplat.h).  vertices [V,3] -> ([P,3], [P,4]), or the cameras of a
    training step at once: vertices [T,V,3] -> ([T,P,3], [T,P,4])."""

    @staticmethod
    def forward(ctx, vertices, bary, rotation, vid, rest, rowptr=None, corners=None):
        ctx.sinks = (_n.grad_sink(bary), _n.grad_sink(rotation))      # (csplat.dist.FlatGrads: write the parameter gradients in place)
        vertices, bary, rotation = vertices.contiguous().float(), bary.contiguous().float(), rotation.contiguous().float()
        batched = vertices.dim() == 3
        T, V = (int(vertices.shape[0]), int(vertices.shape[1])) if batched else (1, int(vertices.shape[0]))
        P = int(vid.shape[0])
        xyz = torch.empty((T, P, 3) if batched else (P, 3), dtype=torch.float32, device=vertices.device)
        quat = torch.empty((T, P, 4) if batched else (P, 4), dtype=torch.float32, device=vertices.device)
        with _n.on_device(vertices.device):
            _n.check(_n.lib.csplat_mesh_transform_fwd_views(_n.stream_handle(vertices.device), T, P, V, _n.ptr(vid),
                                                            _n.ptr(vertices), _n.ptr(bary), _n.ptr(rotation), _n.ptr(rest),
                                                            _n.ptr(xyz), _n.ptr(quat)), "csplat_mesh_transform_fwd_views")
        ctx.save_for_backward(vertices, bary, rotation, vid, rest, rowptr, corners)
        ctx.dims = (T, P, V)
        return xyz, quat

    @staticmethod
    def backward(ctx, g_xyz, g_quat):
        vertices, bary, rotation, vid, rest, rowptr, corners = ctx.saved_tensors
        T, P, V = ctx.dims
        sinks = getattr(ctx, "sinks", (None, None))
        d_v = torch.empty_like(vertices)
        d_b, d_r = _n.grad_out(sinks[0], bary.shape, bary.device), _n.grad_out(sinks[1], rotation.shape, rotation.device)
        scratch = None if rowptr is None else torch.empty(max(T * P * 9, 1), dtype=torch.float32, device=vertices.device)
        g_xyz = None if g_xyz is None else g_xyz.contiguous().float()
        g_quat = None if g_quat is None else g_quat.contiguous().float()
        with _n.on_device(vertices.device):
            _n.check(_n.lib.csplat_mesh_transform_bwd_views(_n.stream_handle(vertices.device), T, P, V, _n.ptr(vid),
                                                            _n.ptr(vertices), _n.ptr(bary), _n.ptr(rotation), _n.ptr(rest),
                                                            _n.ptr(g_xyz), _n.ptr(g_quat), _n.ptr(d_v), _n.ptr(d_b), _n.ptr(d_r),
                                                            _n.ptr(rowptr), _n.ptr(corners), _n.ptr(scratch)),
                     "csplat_mesh_transform_bwd_views")
        return d_v, d_b, d_r, None, None, None, None


def _join_views(grads, rest, dtype, dev):
    """the [T, *rest] gradient of T unbound rows: a VIEW when the T incoming gradients sit back to back in one buffer (the batched
    rasterizer's backward lays the per-view gradients of means3D / rotations out that way, diff_gaussian_rasterization._plan_backward),
    their stack otherwise (None rows = zeros); None when every row is None"""
    if all(g is None for g in grads):
        return None
    numel = 1
    for d in rest:
        numel *= d
    g0 = grads[0]
    if g0 is not None and all(g is not None and g.dtype == g0.dtype and g.is_contiguous() and tuple(g.shape) == tuple(rest) and
                              g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() and
                              g.storage_offset() == g0.storage_offset() + i * numel for i, g in enumerate(grads)):
        strides, acc = [], 1
        for d in reversed(rest):
            strides.append(acc)
            acc *= d
        return g0.as_strided((len(grads),) + tuple(rest), (numel,) + tuple(reversed(strides)))
    return torch.stack([g if g is not None else torch.zeros(rest, dtype=dtype, device=dev) for g in grads], 0)


class MeshTransformViews(torch.autograd.Function):
    """MeshTransform for the T cameras of a step with the per-camera rows as SEPARATE outputs (T means3D [P,3], then T rotations [P,4]):
    one autograd node where MeshTransform + two unbinds are three, and a backward that reads the rasterizer's per-view gradients in place."""

    @staticmethod
    def forward(ctx, vertices, bary, rotation, vid, rest, rowptr=None, corners=None):
        ctx.sinks = (_n.grad_sink(bary), _n.grad_sink(rotation))
        vertices, bary, rotation = vertices.contiguous().float(), bary.contiguous().float(), rotation.contiguous().float()
        T, V, P = int(vertices.shape[0]), int(vertices.shape[1]), int(vid.shape[0])
        xyz = torch.empty(T, P, 3, dtype=torch.float32, device=vertices.device)
        quat = torch.empty(T, P, 4, dtype=torch.float32, device=vertices.device)
        with _n.on_device(vertices.device):
            _n.check(_n.lib.csplat_mesh_transform_fwd_views(_n.stream_handle(vertices.device), T, P, V, _n.ptr(vid),
                                                            _n.ptr(vertices), _n.ptr(bary), _n.ptr(rotation), _n.ptr(rest),
                                                            _n.ptr(xyz), _n.ptr(quat)), "csplat_mesh_transform_fwd_views")
        ctx.save_for_backward(vertices, bary, rotation, vid, rest, rowptr, corners)
        ctx.dims = (T, P, V)
        ctx.set_materialize_grads(False)
        return tuple(xyz.unbind(0)) + tuple(quat.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        T, P, V = ctx.dims
        dev = ctx.saved_tensors[0].device
        g_xyz = _join_views(grads[:T], (P, 3), torch.float32, dev)
        g_quat = _join_views(grads[T:], (P, 4), torch.float32, dev)
        if g_xyz is None and g_quat is None:
            return (None,) * 7
        return MeshTransform.backward(ctx, g_xyz, g_quat)


from .densify import DensifyMixin  # noqa: E402


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_rotation(r):
    """utils/general_utils.py:81-102"""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def _act_fwd(op_raw, sc_raw, f_dc, f_rest):
    P = op_raw.shape[0]
    opacity, scales = torch.empty_like(op_raw), torch.empty_like(sc_raw)
    shs = torch.empty(P, 16, 3, dtype=torch.float32, device=op_raw.device)
    with _n.on_device(op_raw.device):
        _n.check(_n.lib.csplat_gauss_act_fwd(_n.stream_handle(op_raw.device), P, _n.ptr(op_raw), _n.ptr(sc_raw), _n.ptr(f_dc),
                                             _n.ptr(f_rest), _n.ptr(opacity), _n.ptr(scales), _n.ptr(shs)), "csplat_gauss_act_fwd")
    return opacity, scales, shs


def _act_bwd(opacity, scales, g_op, g_sc, g_shs, sinks=(None, None, None, None)):
    P, dev = opacity.shape[0], opacity.device
    c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
    g_op, g_sc, g_shs = c(g_op), c(g_sc), c(g_shs)
    d_op, d_sc = _n.grad_out(sinks[0], opacity.shape, dev), _n.grad_out(sinks[1], scales.shape, dev)
    d_dc = _n.grad_out(sinks[2], (P, 1, 3), dev)
    d_rest = _n.grad_out(sinks[3], (P, 15, 3), dev)
    with _n.on_device(dev):
        _n.check(_n.lib.csplat_gauss_act_bwd(_n.stream_handle(dev), P, _n.ptr(opacity), _n.ptr(scales), _n.ptr(g_op), _n.ptr(g_sc),
                                             _n.ptr(g_shs), _n.ptr(d_op), _n.ptr(d_sc), _n.ptr(d_dc), _n.ptr(d_rest)),
                 "csplat_gauss_act_bwd")
    return d_op, d_sc, d_dc, d_rest


class _GaussianActivations(torch.autograd.Function):
    """sigmoid(_opacity), exp(_scaling), cat(_features_dc, _features_rest): gaussian_model.py:96-121, one HIP launch each way"""

    @staticmethod
    def forward(ctx, op_raw, sc_raw, f_dc, f_rest):
        ctx.sinks = tuple(_n.grad_sink(t) for t in (op_raw, sc_raw, f_dc, f_rest))
        opacity, scales, shs = _act_fwd(op_raw, sc_raw, f_dc, f_rest)
        ctx.save_for_backward(opacity, scales)
        ctx.set_materialize_grads(False)
        return opacity, scales, shs

    @staticmethod
    def backward(ctx, g_op, g_sc, g_shs):
        return _act_bwd(*ctx.saved_tensors, g_op, g_sc, g_shs, sinks=ctx.sinks)


class GaussianStepInputs(torch.autograd.Function):
    """What the rasterizer receives from the Gaussian model for the T cameras of a step, as ONE autograd node: the mesh -> Gaussian
    transform (MeshTransformViews: T rows of means3D, T rows of rotations) and the activations (_GaussianActivations: opacity, scales,
    SH) -- the same two launches each way, one node less to record and to walk (the step is bound by the host's cost per node).
    Outputs: T means3D [P,3], T rotations [P,4], opacity [P,1], scales [P,3], shs [P,16,3]."""

    @staticmethod
    def forward(ctx, vertices, bary, rotation, vid, rest, rowptr, corners, op_raw, sc_raw, f_dc, f_rest):
        ctx.sinks = tuple(_n.grad_sink(t) for t in (bary, rotation, op_raw, sc_raw, f_dc, f_rest))
        vertices, bary, rotation = vertices.contiguous().float(), bary.contiguous().float(), rotation.contiguous().float()
        T, V, P = int(vertices.shape[0]), int(vertices.shape[1]), int(vid.shape[0])
        xyz = torch.empty(T, P, 3, dtype=torch.float32, device=vertices.device)
        quat = torch.empty(T, P, 4, dtype=torch.float32, device=vertices.device)
        with _n.on_device(vertices.device):
            _n.check(_n.lib.csplat_mesh_transform_fwd_views(_n.stream_handle(vertices.device), T, P, V, _n.ptr(vid),
                                                            _n.ptr(vertices), _n.ptr(bary), _n.ptr(rotation), _n.ptr(rest),
                                                            _n.ptr(xyz), _n.ptr(quat)), "csplat_mesh_transform_fwd_views")
        opacity, scales, shs = _act_fwd(op_raw, sc_raw, f_dc, f_rest)
        ctx.save_for_backward(vertices, bary, rotation, vid, rest, rowptr, corners, opacity, scales)
        ctx.dims = (T, P, V)
        ctx.set_materialize_grads(False)
        return tuple(xyz.unbind(0)) + tuple(quat.unbind(0)) + (opacity, scales, shs)

    @staticmethod
    def backward(ctx, *grads):
        T, P, V = ctx.dims
        saved = ctx.saved_tensors
        dev = saved[0].device
        g_xyz = _join_views(grads[:T], (P, 3), torch.float32, dev)
        g_quat = _join_views(grads[T:2 * T], (P, 4), torch.float32, dev)
        d_mesh = (None, None, None)
        if g_xyz is not None or g_quat is not None:
            d_mesh = MeshTransform.backward(_Saved(saved[:7], ctx.dims, ctx.sinks[:2]), g_xyz, g_quat)[:3]
        g_op, g_sc, g_shs = grads[2 * T:]
        d_act = (None,) * 4
        if g_op is not None or g_sc is not None or g_shs is not None:
            d_act = _act_bwd(saved[7], saved[8], g_op, g_sc, g_shs, sinks=ctx.sinks[2:])
        return tuple(d_mesh) + (None, None, None, None) + tuple(d_act)


class _Saved:
    """stand-in ctx: hands MeshTransform.backward its saved tensors and dims"""

    def __init__(self, saved, dims, sinks=(None, None)):
        self.sinks = sinks
        self.saved_tensors, self.dims = saved, dims


class UnbindViews(torch.autograd.Function):
    """x.unbind(0) whose backward does not copy when the T incoming gradients already sit back to back in one buffer (_join_views) --
    where torch's own unbind backward is a stack (one copy launch per tensor and step)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        ctx.meta = (x.dtype, x.device)
        ctx.set_materialize_grads(False)
        return tuple(x.detach().unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        return _join_views(grads, ctx.shape[1:], *ctx.meta)


class MeshGaussians(DensifyMixin):
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.mesh = SimpleNamespace(pos=None, face=None, edge_index=None)
        self.face_ids = torch.empty(0)
        self.face_bary = torch.empty(0)
        self.face_offset = torch.empty(0)
        self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = torch.empty(0)
        self.edge_norm = torch.empty(0)
        self.optimizer = None
        self.fused = True   # mesh -> Gaussian transform through csplat_mesh_transform_* (False: the torch formulation)

    # ---- construction -------------------------------------------------------------------------------------------
    def from_mesh(self, pos, face, edge_index, gaussian_init_factor=2, generator=None):
        """gaussian_mesh.py:207-263: P = factor * F Gaussians, barycentric jitter, scales from distCUDA2."""
        from simple_knn._C import distCUDA2
        require_cuda(pos)
        dev = pos.device
        self.mesh = SimpleNamespace(pos=pos, face=face, edge_index=edge_index)
        disp = pos[edge_index[1]] - pos[edge_index[0]]
        self.edge_norm = disp.norm(dim=-1, keepdim=True)
        F = face.shape[1]
        P = gaussian_init_factor * F
        bary = torch.full((P, 3), 1.0 / 3.0, device=dev)
        if gaussian_init_factor > 1:
            bary = torch.clip(torch.normal(bary, 0.05, generator=generator), 0.0, 1.0)
            bary = bary / bary.sum(dim=1, keepdim=True)
        self.face_bary = nn.Parameter(bary.requires_grad_(True))
        self.face_offset = nn.Parameter(torch.zeros(P, 1, device=dev).requires_grad_(True))
        self.face_ids = torch.arange(0, F, dtype=torch.long, device=dev).repeat(gaussian_init_factor).sort().values
        rgb = torch.rand(P, 3, device=dev, generator=generator) / 255.0
        feats = torch.zeros(P, 3, (self.max_sh_degree + 1) ** 2, device=dev)
        feats[:, :3, 0] = (rgb - 0.5) / 0.28209479177387814
        dist2 = torch.clamp_min(distCUDA2(self.get_xyz().detach()), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros(P, 4, device=dev)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones(P, 1, device=dev))
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opac.requires_grad_(True))
        return self

    def from_arrays(self, pos, face, edge_index, face_ids, bary, log_scales, quats, opacity_logits, sh):
        """load explicit parameters (synthetic scene_1)."""
        self.mesh = SimpleNamespace(pos=pos, face=face, edge_index=edge_index)
        self.edge_norm = (pos[edge_index[1]] - pos[edge_index[0]]).norm(dim=-1, keepdim=True)
        self.face_ids = face_ids
        self.face_bary = nn.Parameter(bary.clone().requires_grad_(True))
        self.face_offset = nn.Parameter(torch.zeros(bary.shape[0], 1, device=bary.device, dtype=bary.dtype).requires_grad_(True))
        self._features_dc = nn.Parameter(sh[:, 0:1].contiguous().clone().requires_grad_(True))
        self._features_rest = nn.Parameter(sh[:, 1:].contiguous().clone().requires_grad_(True))
        self._scaling = nn.Parameter(log_scales.clone().requires_grad_(True))
        self._rotation = nn.Parameter(quats.clone().requires_grad_(True))
        self._opacity = nn.Parameter(opacity_logits.clone().requires_grad_(True))
        return self

    def invalidate_caches(self):
        """drop everything derived from (face_ids, mesh): called by the densification / pruning / loading code"""
        self.__dict__.pop("_rest_cache", None)
        self.__dict__.pop("_fused_cache", None)

    def save_ply(self, path):
        """gaussian_mesh.py:438-465 (point_cloud.ply in plyfile's layout + the mesh side-car), csplat/ply.py"""
        from .ply import save_gaussians
        save_gaussians(self, path)

    def load_ply(self, path, device="cuda"):
        """gaussian_mesh.py:467-481"""
        from .ply import load_gaussians
        return load_gaussians(self, path, device)

    def parameters(self):
        return [self.face_bary, self.face_offset, self._features_dc, self._features_rest, self._opacity, self._scaling,
                self._rotation]

    def training_setup(self, position_lr=1.6e-4, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                       spatial_lr_scale=1.0):
        """the 7 Adam parameter groups of gaussian_mesh.py:126-136"""
        groups = [
            {'params': [self.face_bary], 'lr': position_lr * spatial_lr_scale, "name": "face_bary"},
            {'params': [self.face_offset], 'lr': position_lr * spatial_lr_scale, "name": "face_offset"},
            {'params': [self._features_dc], 'lr': feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': rotation_lr, "name": "rotation"}]
        from .optim import GroupedAdam
        self.optimizer = GroupedAdam(groups, lr=0.0, eps=1e-15)   # torch.optim.Adam semantics and state, one launch per step
        return self.optimizer

    # ---- activations (gaussian_model.py:96-121) -----------------------------------------------------------------
    @property
    def num_gaussians(self):
        return self.face_ids.shape[0]

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def activations(self):
        """(get_opacity, get_scaling, get_features) in one launch each way on the GPU (csplat_gauss_act_fwd/_bwd) -- the three
        tensors render() hands the rasterizer every step; None when the fused form does not apply (CPU, dtype, layout)"""
        ts = (self._opacity, self._scaling, self._features_dc, self._features_rest)
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts) or self._features_rest.shape[1:] != (15, 3) or \
                self._features_dc.shape[1:] != (1, 3) or self._opacity.shape[0] == 0:
            if self._opacity.shape[0]:
                _n.composed_fallback("MeshGaussians.activations", _n.why_not_f32c(*ts) or "shape", *ts)
            return None
        return _GaussianActivations.apply(*ts)

    def get_covariance(self, scaling_modifier=1):
        L = build_rotation(self._rotation) @ torch.diag_embed(scaling_modifier * self.get_scaling)
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- mesh -> Gaussian transform (gaussian_mesh.py:151-188) --------------------------------------------------
    def _vertex_ids(self):
        return self.mesh.face[:, self.face_ids].transpose(0, 1)  # [P, 3]

    def _rest(self):
        """(key, vertex ids [P,3], rest-pose record, vertex<-corner CSR rowptr, corners, ...): what the fused transform needs of the static
        mesh, rebuilt only when face_ids / mesh.pos changed"""
        # keyed on the tensor OBJECTS (held in the cache entry, so their storage cannot be recycled under the key) and their
        # in-place version counters; densify.py also drops the entry whenever it re-creates face_ids
        fi, mp = self.face_ids, self.mesh.pos
        r = self.__dict__.get("_rest_cache")
        key = (fi._version, mp._version, int(fi.shape[0]))
        if r is None or r[0] != key or r[5] is not fi or r[6] is not mp:
            vid = self._vertex_ids().contiguous()
            rest = torch.empty(max(int(_n.lib.csplat_mesh_rest_bytes(vid.shape[0])), 256), dtype=torch.uint8, device=vid.device)
            with _n.on_device(vid.device):
                _n.check(_n.lib.csplat_mesh_rest(_n.stream_handle(vid.device), int(vid.shape[0]), _n.ptr(vid),
                                                 _n.ptr(self.mesh.pos.contiguous().float()), _n.ptr(rest)), "csplat_mesh_rest")
            # vertex <- (Gaussian, corner) incidence, grouped by vertex in ascending pair order: the backward gathers the
            # vertex gradients through it instead of scattering them with atomics (static until the next densification)
            flat = vid.reshape(-1).contiguous()
            nv = int(self.mesh.pos.shape[0])
            rowptr, corners = _n.group_by_key(flat, nv)      # (csplat_gnn_build_csr: counting sort, ascending ids inside a group)
            r = (key, vid, rest, rowptr, corners, fi, mp)
            self._rest_cache = r
        if r[1].shape[0] != self.face_bary.shape[0] or r[1].shape[0] != self._rotation.shape[0]:
            raise _n.CsplatError(f"mesh transform: {r[1].shape[0]} face ids but {self.face_bary.shape[0]} barycentric rows / "
                                 f"{self._rotation.shape[0]} rotations")
        return r

    def _fused(self, deformed_vertices, views=False):
        """(xyz, rotation) on the deformed mesh through the fused HIP kernel; render() asks for both, one after the other
        with the same vertex tensor, so the pair is computed once and cached on that tensor object."""
        c = self.__dict__.get("_fused_cache")
        if c is not None and c[0] is deformed_vertices and c[1] == deformed_vertices._version and c[3] == bool(views):
            return c[2]
        r = self._rest()
        fn = MeshTransformViews if views else MeshTransform       # (views: T rows of means3D, then T rows of rotations)
        out = fn.apply(deformed_vertices, self.face_bary, self._rotation, r[1], r[2], r[3], r[4])
        self._fused_cache = (deformed_vertices, deformed_vertices._version, out, bool(views))
        return out

    def step_inputs(self, deformed_vertices):
        """transform_views + activations as ONE autograd node (GaussianStepInputs): ((T means3D), (T rotations), opacity, scales, shs),
        or None when one of the two fused forms does not apply"""
        ts = (self._opacity, self._scaling, self._features_dc, self._features_rest)
        if not (self.fused and deformed_vertices.is_cuda and deformed_vertices.dim() == 3 and
                all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)) or \
                self._features_rest.shape[1:] != (15, 3) or self._features_dc.shape[1:] != (1, 3) or self._opacity.shape[0] == 0:
            return None
        r = self._rest()
        T = int(deformed_vertices.shape[0])
        out = GaussianStepInputs.apply(deformed_vertices, self.face_bary, self._rotation, r[1], r[2], r[3], r[4], *ts)
        return out[:T], out[T:2 * T], out[2 * T], out[2 * T + 1], out[2 * T + 2]

    def transform_views(self, deformed_vertices):
        """get_xyz + get_rotation for the cameras of a step at once: [T,V,3] -> (tuple of T [P,3], tuple of T [P,4])."""
        T = int(deformed_vertices.shape[0])
        rows = self._fused(deformed_vertices, views=True)
        return rows[:T], rows[T:]

    def get_xyz(self, deformed_vertices=None):
        if deformed_vertices is not None and deformed_vertices.is_cuda and self.fused:
            return self._fused(deformed_vertices)[0]
        vid = self._vertex_ids()
        verts = self.mesh.pos if deformed_vertices is None else deformed_vertices
        face_pos = verts[vid, :]                                                      # [P, 3 (vertex), 3 (xyz)]
        nb = self.face_bary / self.face_bary.sum(dim=1, keepdim=True)
        # (the reference writes this as a [P,1,3] @ [P,3,3] batched matmul; P batched 1x3x3 GEMMs cost ~1 ms each way on
        # hipBLASLt -- the same contraction as a broadcast multiply + sum streams at HBM rate)
        return (nb.unsqueeze(-1) * face_pos).sum(dim=1)

    def vertex_normals(self, vertices):
        """unit vertex normals as torch_geometric.transforms.GenerateMeshNormals defines them (the transform the reference applies,
        gaussian_mesh.py:199-200): unit face normals (v1 - v0) x (v2 - v0), summed onto the face's three vertices, normalised."""
        face = self.mesh.face
        v0, v1, v2 = vertices[face[0]], vertices[face[1]], vertices[face[2]]
        fn = torch.nn.functional.normalize(torch.cross(v1 - v0, v2 - v0, dim=1), p=2, dim=-1)
        vn = torch.zeros_like(vertices).index_add_(0, torch.cat([face[0], face[1], face[2]]), fn.repeat(3, 1))
        return torch.nn.functional.normalize(vn, p=2, dim=-1)

    def get_vertice_rotation(self, deformed_vertices):
        """gaussian_mesh.py:190-201: per-vertex rotation (quaternion, XYZW) from the rest-pose vertex normal to the deformed one --
        axis = n_rest x n_def (normalised), angle = acos(clamp(n_rest . n_def)) (meshnet/data_utils.py:460-491).  A debug by-product:
        render(log_deform_path=...) stores it (gaussian_renderer/__init__.py:118-127); plain torch ops, no kernel."""
        rest = getattr(self.mesh, "norm", None)
        if rest is None:            # (the reference's compute_mesh stores mesh.norm at load time; a mesh built from arrays has none)
            rest = self.vertex_normals(self.mesh.pos)
        deformed = self.vertex_normals(deformed_vertices)
        cross = torch.cross(rest, deformed, dim=1)
        angle = torch.acos(torch.clamp((rest * deformed).sum(dim=1), -1.0, 1.0))
        axis = cross / torch.linalg.norm(cross, dim=1, keepdim=True)
        return torch.cat([axis * torch.sin(angle / 2).unsqueeze(1), torch.cos(angle / 2).unsqueeze(1)], dim=1)

    def get_rotation(self, deformed_vertices=None):
        if deformed_vertices is not None and deformed_vertices.is_cuda and self.fused:
            return self._fused(deformed_vertices)[1]
        rotation = torch.nn.functional.normalize(self._rotation)
        if deformed_vertices is None:
            return rotation
        vid = self._vertex_ids()
        # closed-form 3-point Kabsch (== roma.rigid_points_registration's SVD solution, csplat/rotations.py)
        R = rot.kabsch_triangles(self.mesh.pos[vid, :], deformed_vertices[vid, :])
        return rot.quat_composition([rotation, rot.rotmat_to_unitquat(R)])
