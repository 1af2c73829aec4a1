"""Drop-in for `diff_gaussian_rasterization` (the depth fork), MI355X-native.

Mirrors the interface the reference binds at gaussian_renderer/__init__.py:16 and calls at :61-76,156-164:
    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                                  projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None) -> (color, radii, depth)
Gradients are delivered for means3D, means2D (NDC-space screen gradient used by densification,
scene_reconstruction/train_utils.py:290-292), shs, colors_precomp, opacities, scales, rotations, cov3D_precomp.
The depth image carries no gradient (as upstream).  All compute is in libcsplat.so (csplat_forward_begin / _finish / csplat_backward).
`rasterize_views` renders several independent views in one call, one HIP stream per view.
"""
import contextlib as _contextlib
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


def _f32c(t, device, what="input", per_gaussian=True):
    """fp32, contiguous, on `device`.  The upstream binding takes `.contiguous().data<float>()` of every argument: a non-fp32 tensor
    is REJECTED there and a strided one silently copied.  Here a tensor of another dtype or on another device is converted -- which
    is a torch launch outside the HIP path, so it is reported (csplat.native.composed_fallback: counted, and an error under STRICT,
    kind "dtype"); a strided per-Gaussian tensor is copied like upstream does and reported as kind "layout"; the camera constants
    (4x4 matrices: the reference builds them as `.transpose(0, 1)` views, scene_reconstruction/cameras.py:63-67) are copied
    without a report, exactly upstream's behaviour."""
    if t is None:
        return None
    if t.device != device or t.dtype != torch.float32:
        _n.composed_fallback("diff_gaussian_rasterization." + what, "dtype", t if t.is_cuda else torch.empty(0, device=device))
        t = t.to(device=device, dtype=torch.float32)
    if not t.is_contiguous():
        if per_gaussian:
            _n.composed_fallback("diff_gaussian_rasterization." + what, "layout", t)
        t = t.contiguous()
    return t


def _f32c_grad(t, device):
    """the incoming image gradient: whatever autograd hands over (an expanded scalar, a slice) is made dense -- not the caller's input"""
    if t is None:
        return None
    if t.device != device or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t.contiguous()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """One camera (what GaussianRasterizer.forward calls, gaussian_renderer/__init__.py:156-164).  Since round 5 a single view goes
    through the same library entry as a batch of views (csplat_forward_views_deferred with V = 1): from the second call of an image size
    on, the second forward phase is launched on the previous call's capacities and the counts are read AFTER the host has prepared the
    backward -- a camera-by-camera loop (the reference's train_utils.py:259-272) no longer leaves the GPU idle for a host round trip per
    camera.  Images, radii, depth and gradients are those of _RasterizeGaussians (tests: test_batched_views_equal_single_view_calls)."""
    if PER_CALL_SPECULATION and means3D.is_cuda:
        return _RasterizeGaussiansBatch.apply((raster_settings,), False, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                              cov3Ds_precomp)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


PER_CALL_SPECULATION = True      # False: every call waits for its own counts (csplat_forward_begin / _finish, the path of rounds 1-4)


class _View:
    """Everything one view's forward produces and its backward needs (host side of csplat_forward_begin / _finish)."""

    def __init__(self, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        _n.require_cuda(means3D)
        dev = self.dev = means3D.device
        self.rs = rs
        self.P = int(means3D.shape[0])
        self.H, self.W = int(rs.image_height), int(rs.image_width)
        self.means3D = _f32c(means3D, dev, "means3D"); self.opacities = _f32c(opacities, dev, "opacities")
        self.sh = _f32c(sh, dev, "shs")
        self.colors_precomp = _f32c(colors_precomp, dev, "colors_precomp")
        self.scales = _f32c(scales, dev, "scales")
        self.rotations = _f32c(rotations, dev, "rotations")
        self.cov3Ds_precomp = _f32c(cov3Ds_precomp, dev, "cov3D_precomp")
        self.bg = _f32c(rs.bg, dev, "bg", False); self.view = _f32c(rs.viewmatrix, dev, "viewmatrix", False)
        self.proj = _f32c(rs.projmatrix, dev, "projmatrix", False)
        self.campos = _f32c(rs.campos, dev, "campos", False)
        self.M = int(self.sh.shape[1]) if self.sh is not None else 0
        self.ticket = None

    def inputs(self):
        return [t for t in (self.means3D, self.sh, self.colors_precomp, self.opacities, self.scales, self.rotations,
                            self.cov3Ds_precomp, self.bg, self.view, self.proj, self.campos) if t is not None]

    def begin(self):
        """K1 + the counting half of the binning on the CURRENT stream; nothing here waits for the GPU."""
        dev, rs = self.dev, self.rs
        self.color = torch.empty(3, self.H, self.W, dtype=torch.float32, device=dev)
        self.depth = torch.empty(1, self.H, self.W, dtype=torch.float32, device=dev)
        self.radii = torch.empty(self.P, dtype=torch.int32, device=dev)
        self.alloc = _n.ChunkAllocator(dev)
        tk = C.c_int(-1)
        with _n.on_device(dev):
            rc = _n.lib.csplat_forward_begin(
                _n.stream_handle(dev), self.P, int(rs.sh_degree), self.M, _n.ptr(self.bg), self.W, self.H,
                _n.ptr(self.means3D), _n.ptr(self.sh), _n.ptr(self.colors_precomp), _n.ptr(self.opacities),
                _n.ptr(self.scales), float(rs.scale_modifier), _n.ptr(self.rotations), _n.ptr(self.cov3Ds_precomp),
                _n.ptr(self.view), _n.ptr(self.proj), _n.ptr(self.campos), float(rs.tanfovx), float(rs.tanfovy),
                int(bool(rs.prefiltered)), self.alloc.cb, None, _n.ptr(self.radii), C.byref(tk))
        _n.check(rc, "csplat_forward_begin")
        self.ticket = int(tk.value)

    def finish(self):
        """reads num_rendered (the one host round trip), allocates the R-sized chunks, K3..K6 on the begin() stream"""
        R = C.c_int(0)
        geom, binning, image = C.c_void_p(), C.c_void_p(), C.c_void_p()
        with _n.on_device(self.dev):
            rc = _n.lib.csplat_forward_finish(self.ticket, _n.ptr(self.color), _n.ptr(self.depth), C.byref(R),
                                              C.byref(geom), C.byref(binning), C.byref(image))
        self.ticket = None
        _n.check(rc, "csplat_forward_finish")
        self.num_rendered = int(R.value)
        ch = self.alloc.chunks
        self.chunks = (ch[_n_GEOM], ch[_n_BINNING], ch[_n_IMAGE])
        self.alloc.cb = None  # break the allocator <-> callback cycle: TEMP / TABLE die here, not at the next cyclic GC
        self.alloc = None

    def backward(self, grad_color, saved):
        """K7 + K8 on the CURRENT stream; returns the per-input gradients in the Function's argument order."""
        means3D, sh, colors_precomp, scales, rotations, cov3Ds_precomp, radii, color = saved
        rs, dev, P, M = self.rs, self.dev, self.P, self.M
        geom, binning, image = self.chunks
        grad_color = _f32c_grad(grad_color, dev)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        d_mean2D, d_conic, d_opac, d_color = new(P, 3), new(P, 4), new(P, 1), new(P, 3)
        d_mean3D, d_cov3D = new(P, 3), new(P, 6)
        d_sh = new(P, M, 3) if sh is not None else None
        d_scale = new(P, 3) if scales is not None else None
        d_rot = new(P, 4) if rotations is not None else None
        scratch = torch.empty(max(int(_n.lib.csplat_backward_scratch_bytes(P, self.num_rendered)), 256), dtype=torch.uint8,
                              device=dev)
        with _n.on_device(dev):
            rc = _n.lib.csplat_backward(
                _n.stream_handle(dev), P, int(rs.sh_degree), M, self.num_rendered, _n.ptr(self.bg), self.W, self.H,
                _n.ptr(means3D), _n.ptr(sh), _n.ptr(colors_precomp), _n.ptr(scales), float(rs.scale_modifier),
                _n.ptr(rotations), _n.ptr(cov3Ds_precomp), _n.ptr(self.view), _n.ptr(self.proj), _n.ptr(self.campos),
                float(rs.tanfovx), float(rs.tanfovy), _n.ptr(radii), _n.ptr(geom), _n.ptr(binning), _n.ptr(image),
                _n.ptr(color), _n.ptr(grad_color), _n.ptr(scratch), _n.ptr(d_mean2D), _n.ptr(d_conic), _n.ptr(d_opac),
                _n.ptr(d_color), _n.ptr(d_mean3D), _n.ptr(d_cov3D), _n.ptr(d_sh), _n.ptr(d_scale), _n.ptr(d_rot))
        _n.check(rc, "csplat_backward")
        return (d_mean3D, d_mean2D, d_sh, d_color if colors_precomp is not None else None, d_opac, d_scale, d_rot,
                d_cov3D if cov3Ds_precomp is not None else None)

    def saved(self):
        return (self.means3D, self.sh, self.colors_precomp, self.scales, self.rotations, self.cov3Ds_precomp, self.radii,
                self.color)

    def drop_inputs(self):
        """the tensors travel through ctx.save_for_backward; keep only constants and chunks here"""
        self.means3D = self.sh = self.colors_precomp = self.opacities = self.scales = self.rotations = None
        self.cov3Ds_precomp = self.radii = self.color = self.depth = None


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        v = _View(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs)
        v.begin()
        v.finish()
        color, radii, depth = v.color, v.radii, v.depth
        ctx.save_for_backward(*v.saved())
        v.drop_inputs()
        ctx.view_state = v
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth):
        return ctx.view_state.backward(grad_color, ctx.saved_tensors) + (None,)


_side_streams = {}
# K7's per-Gaussian accumulation records, kept from step to step (round 5): zeroed ONCE per (device, stream the work runs on, P, view slot);
# K8 clears every record it consumes (csplat.h: CSPLAT_SCRATCH_ZEROED), so the next step finds them zero again -- no clearing launch and
# no 25.6 MB of zero fill per step.  Launches on one stream cannot overlap, so two backward passes that share a slot are ordered.  A failed
# entry point may leave records behind: the cache is a ticket cache (csplat.native.check drops it, recordings are invalidated).
_ACC_SCRATCH = {}
_n.TICKET_CACHES.append(_ACC_SCRATCH)
SCRATCH_ZEROED = 256       # csplat.h: CSPLAT_SCRATCH_ZEROED


def _acc_scratch(dev, P, slot):
    """(tensor, True) = the persistent zeroed records of `slot`; (None, False) in the bit-reproducible mode (its scratch is laid out per call)"""
    if int(_n.lib.csplat_debug_flags_query()) & 256:
        return None, False
    key = (dev.index, _n.scratch_stream(dev), int(P), int(slot))
    buf = _ACC_SCRATCH.get(key)
    if buf is None:
        if len(_ACC_SCRATCH) >= 32:
            _n.evict_scratch(_ACC_SCRATCH)
        buf = _ACC_SCRATCH[key] = torch.zeros(max(int(_n.lib.csplat_backward_scratch_bytes(int(P), 0)), 256), dtype=torch.uint8, device=dev)
    return buf, True
# how the batched forward's second phase went, per call (bench.py's `speculation` field, tests): "hit" = launched on the previous call's
# capacities before the counts were read and the counts fitted; "miss" = they did not fit and the phase was repeated with exact sizes;
# "wait" = no history for this (image size, P) yet (or speculation off): the counts were read first
SPEC_STATS = {"hit": 0, "miss": 0, "wait": 0}
# the device words (int32 [3] views into the IMAGE chunks: tile instances, longest tile list, non-empty tiles) of the last batched
# forward, per view -- what a caller sizes a launch on faith from (one .cpu() read at a moment of its choosing)
LAST_INFO = []
# ---- launch mode of the batched forward.  Two switches, both PRIVATE and both set only through `forward_mode(...)` below, which restores
# them on exit whatever happens inside (module globals toggled around a capture leak silently -- one leaked flag and the
# oracle-tested path and the timed path diverge):
#   keep_info  the call leaves LAST_INFO (the views cost ~10 us of host time per call)
#   faith      launch ON FAITH (csplat.train.CapturedStep, csplat.graphs.ReplayedSteps): a dict {"caps": (R, L, B), "valid": uint32 tensor
#              [1]} -- the batched forward launches both phases with those capacities and reads NOTHING back
#              (csplat_forward_views_faith): the form a hipGraph capture can record.  The forward leaves in the dict: "info" = per view an
#              int32 [3] VIEW of the device words holding (tile instances, longest tile list, non-empty tiles), to be read whenever the
#              caller syncs anyway.
_KEEP_INFO = False
_FAITH = None


@_contextlib.contextmanager
def forward_mode(faith=None, keep_info=False, replay_device=None):
    """Scope in which the batched forward runs in another launch mode: `faith` / `keep_info` as above; `replay_device` = a device whose
    CURRENT stream is the one the recording made inside this scope will be replayed on (scratch buffers that are "zeroed once per
    stream" are keyed on it instead of on the capture stream, csplat.native.scratch_stream).  Not re-entrant for `faith` /
    `replay_device`: nesting two recordings is a bug and raises."""
    global _FAITH, _KEEP_INFO
    if faith is not None and _FAITH is not None:
        raise RuntimeError("diff_gaussian_rasterization.forward_mode: a launch on faith is already in progress")
    prev = (_FAITH, _KEEP_INFO)
    idx = None
    if replay_device is not None:
        idx = replay_device.index if isinstance(replay_device, torch.device) else int(replay_device)
        idx = torch.cuda.current_device() if idx is None else idx
        if idx in _n.REPLAY_STREAM:
            raise RuntimeError("diff_gaussian_rasterization.forward_mode: a recording for this device is already in progress")
        _n.REPLAY_STREAM[idx] = _n.stream_handle(idx)
    _FAITH = faith if faith is not None else prev[0]
    _KEEP_INFO = bool(keep_info) or prev[1]
    try:
        yield faith
    finally:
        _FAITH, _KEEP_INFO = prev
        if idx is not None:
            _n.REPLAY_STREAM.pop(idx, None)


class DeferredK8:
    """Filled by `deferred_k8()`: the batched backward passes inside the scope launched their K7 only; `launch(slice, nslices)` launches
    the per-Gaussian backward (K8) of one Gaussian range for all of them, `rows(slice, nslices)` says which gradient rows that finishes
    (csplat_backward_views_parts / csplat_backward_slice_rows).  A view-parallel step hands the finished rows to its collective while
    the next slice computes (csplat.dist.FlatGrads.start_ranges).  The entries hold raw pointers of the pass that made them: valid as
    long as its gradient tensors are (for a recorded step: as long as the recording)."""

    def __init__(self):
        self.entries = []           # (csplat_view array, number of views, device, P, keep-alive)

    def launch(self, slice_, nslices):
        for sub, n, dev, _P, _keep in self.entries:
            with _n.on_device(dev):
                rc = _n.lib.csplat_backward_views_parts(n, C.cast(sub, C.c_void_p), torch.cuda.current_stream(dev).cuda_stream, 2, int(slice_),
                                                        int(nslices))
            _n.check(rc, "csplat_backward_views_parts")

    def rows(self, slice_, nslices):
        P = self.entries[0][3]
        lo, hi = C.c_int64(0), C.c_int64(0)
        _n.check(_n.lib.csplat_backward_slice_rows(int(P), int(slice_), int(nslices), C.byref(lo), C.byref(hi)), "csplat_backward_slice_rows")
        return int(lo.value), int(hi.value)


_K8_DEFER = None


@_contextlib.contextmanager
def deferred_k8(holder=None):
    """Scope in which `_RasterizeGaussiansBatch.backward` launches the compositing backward (K7) only and leaves the per-Gaussian backward
    (K8) to the caller: `with deferred_k8() as h: loss.backward()`, then `h.launch(g, G)` for g = 0 .. G-1.  The gradient tensors autograd
    has been handed are filled by those launches -- nothing may read them in between: every input of the node must be a LEAF (or feed
    leaves through nodes that do not read values, like views), each receiving its gradient from this node alone."""
    global _K8_DEFER
    if _K8_DEFER is not None:
        raise RuntimeError("diff_gaussian_rasterization.deferred_k8: already inside a deferred_k8 scope")
    _K8_DEFER = holder if holder is not None else DeferredK8()
    try:
        yield _K8_DEFER
    finally:
        _K8_DEFER = None


def forward_mode_is_default():
    """True when no forward_mode scope is open (tests assert it after every recording)"""
    return _FAITH is None and not _KEEP_INFO and not _n.REPLAY_STREAM


def _view_streams(dev, n, main):
    """the caller's stream + n-1 side streams: ROCm maps streams onto 4 hardware queues per device by default, and a
    fifth stream shares a queue with (= is serialised behind) another one"""
    pool = _side_streams.setdefault((dev.type, dev.index), [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=dev))
    return [main] + pool[:n - 1]


class _RasterizeGaussiansBatch(torch.autograd.Function):
    """V independent views in one autograd node and ONE library call each way (csplat_forward_views /
    csplat_backward_views).  Every view runs on its own HIP stream (the caller's + V-1 side streams, fenced inside the
    library): all K1/K2 are issued before the first num_rendered read, the under-filled compositing kernels of the views
    overlap, and a parameter tensor passed to several views gets ONE gradient buffer that the views' K8 add into.
    Images / radii / depth are bit-identical to V calls of _RasterizeGaussians."""

    NIN = 8
    # slot in the per-view argument list -> (csplat_view gradient field, accumulate bit)
    _GRAD = {0: ("dL_dmean3D", _n.ACC_MEAN3D), 2: ("dL_dsh", _n.ACC_SH), 3: ("dL_dcolor", _n.ACC_COLOR),
             4: ("dL_dopacity", _n.ACC_OPACITY), 5: ("dL_dscale", _n.ACC_SCALE), 6: ("dL_drot", _n.ACC_ROT),
             7: ("dL_dcov3D", _n.ACC_COV3D)}

    @staticmethod
    def forward(ctx, settings, stacked, *flat):
        V, n = len(settings), _RasterizeGaussiansBatch.NIN
        assert len(flat) == V * n
        views = []
        for i in range(V):
            means3D, _means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds = flat[i * n:(i + 1) * n]
            views.append(_View(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds, settings[i]))
        dev = views[0].dev
        main = torch.cuda.current_stream(dev)
        streams = _view_streams(dev, V, main)
        arr = (_n.CsplatView * V)()
        chunks = [dict() for _ in range(V)]
        # stacked: the V images land in ONE [V, 3, H, W] tensor (what the reference builds with torch.cat before the loss,
        # scene_reconstruction/train_utils.py:262-270) and that tensor is the node's first output
        if stacked:
            assert all((v.H, v.W) == (views[0].H, views[0].W) for v in views), "stacked output needs equal image sizes"
            colors = torch.empty(V, 3, views[0].H, views[0].W, dtype=torch.float32, device=dev)

        def _alloc(ctx_, chunk, nbytes):     # all on the caller's stream: the library fences the view streams around it
            try:
                buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
                chunks[int(ctx_ or 0)][int(chunk)] = buf
                return buf.data_ptr()
            except Exception:
                return None
        cb = _n.ALLOC_FN(_alloc)
        for i, (v, st) in enumerate(zip(views, streams)):
            rs, w = v.rs, arr[i]
            v.color = colors[i] if stacked else torch.empty(3, v.H, v.W, dtype=torch.float32, device=dev)
            v.depth = torch.empty(1, v.H, v.W, dtype=torch.float32, device=dev)
            v.radii = torch.empty(v.P, dtype=torch.int32, device=dev)
            w.stream = st.cuda_stream
            w.P, w.D, w.M, w.W, w.H, w.prefiltered = v.P, int(rs.sh_degree), v.M, v.W, v.H, int(bool(rs.prefiltered))
            w.scale_modifier, w.tanfovx, w.tanfovy = float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy)
            w.bg, w.means3D, w.shs, w.colors_precomp = _n.ptr(v.bg), _n.ptr(v.means3D), _n.ptr(v.sh), _n.ptr(v.colors_precomp)
            w.opacities, w.scales, w.rotations = _n.ptr(v.opacities), _n.ptr(v.scales), _n.ptr(v.rotations)
            w.cov3D_precomp, w.view, w.proj, w.campos = _n.ptr(v.cov3Ds_precomp), _n.ptr(v.view), _n.ptr(v.proj), _n.ptr(v.campos)
            w.alloc_ctx = i
            w.out_color, w.out_depth, w.radii = _n.ptr(v.color), _n.ptr(v.depth), _n.ptr(v.radii)
        # the one host read of the call (its counts) is DEFERRED when the library can launch the second phase on the previous call's
        # capacities: everything below that does not need the counts -- the output lists, the backward plan -- is host work done
        # while the GPU runs K1..K6, instead of after a ~45 us wait for K1 / K2
        pending = C.c_int(0)
        faith = _FAITH
        if faith is not None:
            caps = (C.c_uint32 * 3)(*[int(c) for c in faith["caps"]])
            with _n.on_device(dev):
                rc = _n.lib.csplat_forward_views_faith(V, C.cast(arr, C.c_void_p), cb, main.cuda_stream, C.cast(caps, C.c_void_p),
                                                       _n.ptr(faith["valid"]))
            _n.check(rc, "csplat_forward_views_faith")
            off = int(_n.lib.csplat_image_info_offset(views[0].W, views[0].H))
            faith["info"] = [chunks[i][_n_IMAGE][off:off + 12].view(torch.int32) for i in range(V)]
        else:
            with _n.on_device(dev):
                rc = _n.lib.csplat_forward_views_deferred(V, C.cast(arr, C.c_void_p), cb, main.cuda_stream, C.byref(pending))
            _n.check(rc, "csplat_forward_views_deferred")
        try:
            return _RasterizeGaussiansBatch._finish_forward(ctx, views, arr, chunks, flat, stacked, colors if stacked else None, dev, main,
                                                            pending, cb)
        except BaseException:
            if pending.value:        # never leave a call pending in the library (its tickets stay reserved otherwise)
                dummy = C.c_int(0)
                try:
                    with _n.on_device(dev):
                        _n.lib.csplat_forward_views_settle(V, C.cast(arr, C.c_void_p), main.cuda_stream, C.byref(dummy))
                except Exception:
                    pass
            raise

    @staticmethod
    def _finish_forward(ctx, views, arr, chunks, flat, stacked, colors, dev, main, pending, cb):
        V, n = len(views), _RasterizeGaussiansBatch.NIN
        outs, saved = [], []
        for i, v in enumerate(views):
            v.num_rendered = int(arr[i].num_rendered)            # (-1 while the call is pending)
            v.layout_rendered = int(arr[i].layout_rendered)      # >= num_rendered: what the binning chunk was laid out for
            v.chunks = (chunks[i][_n_GEOM], chunks[i][_n_BINNING], chunks[i][_n_IMAGE])
            outs += [v.radii, v.depth] if stacked else [v.color, v.radii, v.depth]
            saved += list(v.saved()[:-1]) + ([] if stacked else [v.color])
            ctx.mark_non_differentiable(v.radii, v.depth)
        if stacked:
            outs = [colors] + outs
            saved.append(colors)
        ctx.stacked = bool(stacked)
        ctx.nsaved = len(views[0].saved()) - (1 if stacked else 0)
        ctx.save_for_backward(*saved)
        # which view first received each input tensor OBJECT (shared parameters get one gradient buffer)
        ctx.first_of = [[next(j for j in range(i + 1) if flat[j * n + k] is flat[i * n + k]) for k in range(n)] for i in range(V)]
        for v in views:
            v.drop_inputs()
        ctx.views, ctx.arr = views, arr
        ctx.set_materialize_grads(False)     # an unused view arrives as None in backward() and costs nothing
        ctx.plan = None
        if any(ctx.needs_input_grad):
            # the GPU is busy with K1..K6 of the views right now: prepare the backward call in its shadow
            ctx.plan = _RasterizeGaussiansBatch._plan_backward(views, saved, ctx.nsaved, arr, ctx.first_of, list(range(V)), dev, flat if _n.GRAD_SINK else None)
        if not pending.value and _FAITH is None:
            SPEC_STATS["wait"] += 1
        if pending.value:
            relaunched = C.c_int(0)
            with _n.on_device(dev):
                rc = _n.lib.csplat_forward_views_settle(V, C.cast(arr, C.c_void_p), main.cuda_stream, C.byref(relaunched))
            _n.check(rc, "csplat_forward_views_settle")
            for i, v in enumerate(views):
                v.num_rendered = int(arr[i].num_rendered)
            pending.value = 0
            SPEC_STATS["miss" if relaunched.value else "hit"] += 1
            if relaunched.value:     # the speculation missed: new binning chunks, new layout -> the plan is rebuilt (rare)
                for i, v in enumerate(views):
                    v.layout_rendered = int(arr[i].layout_rendered)
                    v.chunks = (chunks[i][_n_GEOM], chunks[i][_n_BINNING], chunks[i][_n_IMAGE])
                if ctx.plan is not None:
                    _n.grad_release(ctx.plan["sinks"])
                    ctx.plan = _RasterizeGaussiansBatch._plan_backward(views, saved, ctx.nsaved, arr, ctx.first_of, list(range(V)), dev, flat if _n.GRAD_SINK else None)
            elif ctx.plan is not None:
                for a, i in enumerate(ctx.plan["active"]):
                    ctx.plan["sub"][a].num_rendered = arr[i].num_rendered
                    ctx.plan["sub"][a].busy_tiles = arr[i].busy_tiles
        if _KEEP_INFO and _FAITH is None and len({(v.W, v.H) for v in views}) == 1 and all(_n_IMAGE in c for c in chunks):
            off = int(_n.lib.csplat_image_info_offset(views[0].W, views[0].H))
            LAST_INFO[:] = [chunks[i][_n_IMAGE][off:off + 12].view(torch.int32) for i in range(V)]
        return tuple(outs)

    @staticmethod
    def _plan_backward(views, saved, k, arr, first_of, active, dev, inputs=None):
        """Everything of the backward call that does not depend on the incoming gradient values: ONE allocation for every
        gradient, temporary and K7 record of the step (returned gradients are views of it), the csplat_view array with all
        pointers but dL_dpix, the accumulate masks of shared parameters.  Built at the end of forward(), while the
        compositing kernels run, so that backward() is left with pointer patching and one library call.
        inputs: the call's input tensors -- one with a registered gradient SINK (csplat.native.GRAD_SINK: a view-parallel step's flat
        buffer) gets its gradient written there by K8 and handed to autograd as a fresh view of the sink."""
        n = _RasterizeGaussiansBatch.NIN
        plan, owner, total = [], {}, 0
        sink_of = {}        # (view, slot) -> a fresh view of the input's gradient sink
        used_sinks = []

        def reserve(numel):
            nonlocal total
            off = total
            total += (int(numel) + 63) & ~63          # 256-byte granules
            return off
        # per-view gradients of means3D (slot 0) / rotations (slot 6) that belong to DISTINCT input tensors of equal size -- the rows of
        # one [T, P, .] tensor upstream (csplat.gaussians.UnbindViews) -- are laid out back to back, so that their stack is a view
        pre = {}
        for slot, width in ((0, 3), (6, 4)):
            if len(active) > 1 and all(first_of[i][slot] == i for i in active) and len({views[i].P for i in active}) == 1 and \
                    all(saved[i * k + (0 if slot == 0 else 4)] is not None for i in active):
                blk = reserve(len(active) * width * views[active[0]].P)
                for a, i in enumerate(active):
                    pre[(i, slot)] = blk + a * width * views[i].P
        for i in active:
            v = views[i]
            means3D, sh, colors_precomp, scales, rotations, cov3Ds, radii = saved[i * k:i * k + 7]
            P, M = v.P, v.M
            acc_buf, zeroed = _acc_scratch(dev, P, len(plan))
            ent = {"scratch": acc_buf if zeroed else reserve(int(_n.lib.csplat_backward_scratch_bytes(P, v.layout_rendered)) // 4 + 64),
                   "dL_dmean2D": reserve(3 * P), "dL_dconic": reserve(4 * P), "mask": SCRATCH_ZEROED if zeroed else 0, "ret": {}}
            ent["ret"][1] = (ent["dL_dmean2D"], (P, 3))
            shapes = {0: (P, 3), 2: (P, M, 3) if sh is not None else None, 3: (P, 3), 4: (P, 1),
                      5: (P, 3) if scales is not None else None, 6: (P, 4) if rotations is not None else None, 7: (P, 6)}
            present = {0: True, 2: sh is not None, 3: colors_precomp is not None, 4: True, 5: scales is not None,
                       6: rotations is not None, 7: cov3Ds is not None}
            for slot, (field, bit) in _RasterizeGaussiansBatch._GRAD.items():
                if shapes[slot] is None:
                    ent[field] = None
                    continue
                j = first_of[i][slot]
                if present[slot] and j != i and j in active and (slot != 2 or M == 16):
                    ent[field] = owner[(j, slot)]
                    ent["mask"] |= bit
                else:
                    numel = 1
                    for d in shapes[slot]:
                        numel *= d
                    sk = _n.grad_sink(inputs[i * n + slot]) if (inputs is not None and present[slot] and (i, slot) not in pre and
                                                                inputs[i * n + slot] is not None) else None
                    sv = _n.grad_out(sk, inputs[i * n + slot].shape, dev) if sk is not None else None
                    if sv is not None and sv.numel() == numel and sv.data_ptr() % 16 == 0:
                        sink_of[(i, slot)] = sv
                        # (only a sink this plan really CLAIMED is given back when the plan is rebuilt: grad_out() hands out fresh memory
                        # when another node holds the slice, and releasing that node's claim would let a third writer alias it -- ADVICE r5)
                        if sv.data_ptr() == sk[0].data_ptr() + 4 * int(sk[1]):
                            used_sinks.append(sk)
                        ent[field] = owner[(i, slot)] = ("sink", i, slot)
                        ent["ret"][slot] = (ent[field], shapes[slot])
                        continue
                    ent[field] = owner[(i, slot)] = pre[(i, slot)] if (i, slot) in pre else reserve(numel)
                    if present[slot]:
                        ent["ret"][slot] = (ent[field], shapes[slot])
            plan.append(ent)
        big = torch.empty(max(total, 64), dtype=torch.float32, device=dev)
        base = big.data_ptr()
        sub = (_n.CsplatView * len(active))()
        out = [None] * (len(views) * n)
        for a, i in enumerate(active):
            ent = plan[a]
            C.memmove(C.byref(sub[a]), C.byref(arr[i]), C.sizeof(_n.CsplatView))
            w = sub[a]
            w.scratch = ent["scratch"].data_ptr() if torch.is_tensor(ent["scratch"]) else base + 4 * ent["scratch"]
            w.accmask = ent["mask"]
            for field in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D", "dL_dsh",
                          "dL_dscale", "dL_drot"):
                off = ent.get(field)
                if isinstance(off, tuple):
                    setattr(w, field, sink_of[off[1:]].data_ptr())
                else:
                    setattr(w, field, None if off is None else base + 4 * off)
            for slot, (off, shape) in ent["ret"].items():
                if isinstance(off, tuple):
                    out[i * n + slot] = sink_of[off[1:]]
                    continue
                numel = 1
                for d in shape:
                    numel *= d
                out[i * n + slot] = big[off:off + numel].view(shape)
        return {"active": list(active), "big": big, "sub": sub, "out": out, "sinks": used_sinks,
                "acc": [e["scratch"] for e in plan if torch.is_tensor(e["scratch"])]}      # (kept alive with the plan)

    @staticmethod
    def backward(ctx, *grads):
        views, k, arr = ctx.views, ctx.nsaved, ctx.arr
        V = len(views)
        dev = views[0].dev
        main = torch.cuda.current_stream(dev)
        if ctx.stacked:
            gcol = [None] * V if grads[0] is None else [grads[0][i] for i in range(V)]
        else:
            gcol = [grads[3 * i] for i in range(V)]
        active = [i for i in range(V) if gcol[i] is not None]
        if not active:
            return (None, None) + (None,) * (V * _RasterizeGaussiansBatch.NIN)
        plan, ctx.plan = ctx.plan, None                         # (one use: the buffers are handed to autograd)
        if plan is not None and plan["acc"] and (int(_n.lib.csplat_debug_flags_query()) & 256):
            _n.grad_release(plan["sinks"])                      # (the bit-reproducible mode was switched on after the forward: its scratch
            plan = None                                         #  has another layout -- plan again)
        if plan is None or plan["active"] != active:            # a view's image went unused, or a second backward pass
            if plan is not None:
                _n.grad_release(plan["sinks"])
            plan = _RasterizeGaussiansBatch._plan_backward(views, ctx.saved_tensors, k, arr, ctx.first_of, active, dev)
        gs = [_f32c_grad(gcol[i], dev) for i in active]
        for a, g in enumerate(gs):
            plan["sub"][a].dL_dpix = _n.ptr(g)
        if _K8_DEFER is not None:
            # K7 now, K8 in slices later (DeferredK8.launch): the plan -- the view array and every buffer it points into -- stays alive
            with _n.on_device(dev):
                rc = _n.lib.csplat_backward_views_parts(len(active), C.cast(plan["sub"], C.c_void_p), main.cuda_stream, 1, 0, 1)
            _n.check(rc, "csplat_backward_views_parts")
            # (kept alive: the MEMORY the K8 slices write -- never the gradient tensor objects themselves: AccumulateGrad adopts a
            #  gradient only when nobody else holds it, and would otherwise snapshot the still unwritten buffer into a copy)
            outs, plan["out"] = tuple(plan["out"]), None
            _K8_DEFER.entries.append((plan["sub"], len(active), dev, views[active[0]].P, (plan["big"], plan["acc"], gs, ctx.saved_tensors)))
            return (None, None) + outs
        with _n.on_device(dev):
            rc = _n.lib.csplat_backward_views(len(active), C.cast(plan["sub"], C.c_void_p), main.cuda_stream)
        _n.check(rc, "csplat_backward_views")
        return (None, None) + tuple(plan["out"])


def rasterize_views(settings, inputs, stacked=False):
    """Batched entry (no counterpart upstream, where cameras are rendered one by one in a Python loop --
    scene_reconstruction/train_utils.py:204-260): `settings` a list of GaussianRasterizationSettings, `inputs` a list of
    dicts with the keyword names of GaussianRasterizer.forward.  Returns a list of (color, radii, depth); with
    stacked=True (equal image sizes) returns (colors [V,3,H,W], [(colors[i], radii, depth), ...]) where `colors` is the
    differentiable output -- the batch the reference assembles with torch.cat before its losses -- and colors[i] are
    plain slices of it."""
    flat = []
    for kw in inputs:
        shs, cp = kw.get("shs"), kw.get("colors_precomp")
        sc, ro, cov = kw.get("scales"), kw.get("rotations"), kw.get("cov3D_precomp")
        if (shs is None) == (cp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((sc is None or ro is None) and cov is None) or ((sc is not None or ro is not None) and cov is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        flat += [kw["means3D"], kw["means2D"], shs, cp, kw["opacities"], sc, ro, cov]
    res = _RasterizeGaussiansBatch.apply(tuple(settings), bool(stacked), *flat)
    if stacked:
        colors = res[0]
        return colors, [(colors[i], res[1 + 2 * i], res[2 + 2 * i]) for i in range(len(settings))]
    return [tuple(res[3 * i:3 * i + 3]) for i in range(len(settings))]


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
