"""ctypes binding of libcsplat.so (C-ABI declared in include/csplat.h).  torch is used only for device
memory and streams; no torch type crosses the ABI (data_ptr() integers and the raw hipStream_t do)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSPLAT_LIB=<path>: load another build of the library (same-box A/B of kernel variants: tools/ab_libs.sh) -- the shipped file is never
# overwritten by an experiment
LIB_PATH = os.environ.get("CSPLAT_LIB") or os.path.join(_HERE, "libcsplat.so")
ABI_VERSION = 6

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension has not been built.  Run `python __graft_entry__.py build` "
        "(or cloth-splatting_amd/csrc/build.sh).  There is deliberately no fallback path.")

lib = C.CDLL(LIB_PATH)

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)
_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

EXPORTS = {
    "csplat_abi_version": (_i, []),
    "csplat_last_error": (C.c_char_p, []),
    "csplat_debug_flags": (_i, [C.c_uint]),
    "csplat_debug_flags_query": (C.c_uint, []),
    "csplat_debug_stamps": (_i, [_vp, _sz]),
    "csplat_geom_bytes": (_sz, [_i]),
    "csplat_image_bytes": (_sz, [_i, _i]),
    "csplat_binning_bytes": (_sz, [_i64, _i, _i]),
    "csplat_temp_bytes": (_sz, [_i, _i64, _i, _i]),
    "csplat_backward_scratch_bytes": (_sz, [_i, _i64]),
    "csplat_geom_layout": (_i, [_i, C.POINTER(_sz)]),
    "csplat_binning_layout": (_i, [_i64, _i, _i, C.POINTER(_sz)]),
    "csplat_binning_fields": (_i, [_i64, _i, _i, C.POINTER(_sz)]),
    "csplat_image_layout": (_i, [_i, _i, C.POINTER(_sz)]),
    "csplat_forward": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                            _i, ALLOC_FN, _vp, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp),
                            C.POINTER(_vp)]),
    "csplat_forward_begin": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                  _i, ALLOC_FN, _vp, _vp, C.POINTER(_i)]),
    "csplat_forward_finish": (_i, [_i, _vp, _vp, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "csplat_forward_views": (_i, [_i, _vp, ALLOC_FN, _vp]),
    "csplat_forward_views_deferred": (_i, [_i, _vp, ALLOC_FN, _vp, _vp]),
    "csplat_forward_views_settle": (_i, [_i, _vp, _vp, _vp]),
    "csplat_forward_views_faith": (_i, [_i, _vp, ALLOC_FN, _vp, _vp, _vp]),
    "csplat_image_info_offset": (C.c_size_t, [_i, _i]),
    "csplat_backward_views": (_i, [_i, _vp, _vp]),
    "csplat_backward_views_parts": (_i, [_i, _vp, _vp, C.c_uint, _i, _i]),
    "csplat_backward_slice_rows": (_i, [_i, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "csplat_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_dist2": (_i, [_vp, _i, _vp, _vp]),
    "csplat_dist2_temp_bytes": (_sz, [_i]),
    "csplat_dist2_ws": (_i, [_vp, _i, _vp, _vp, _vp]),
    "csplat_mesh_rest_bytes": (_sz, [_i]),
    "csplat_mesh_rest": (_i, [_vp, _i, _vp, _vp, _vp]),
    "csplat_mesh_transform_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_mesh_transform_bwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_mesh_transform_fwd_views": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_mesh_transform_bwd_views": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_blur11": (_i, [_vp, _i64, _i, _i, C.POINTER(C.c_float), _vp, _vp]),
    "csplat_ssim_partial_count": (_sz, [_i64, _i, _i]),
    "csplat_ssim_fwd": (_i, [_vp, _i64, _i, _i, C.POINTER(C.c_float), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_ssim_bwd": (_i, [_vp, _i64, _i, _i, C.POINTER(C.c_float), _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "csplat_adam_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _i64]),
    "csplat_gather_words": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "csplat_adam_step_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "csplat_l1_scratch_bytes": (_sz, []),
    "csplat_mask_to_map_temp_bytes": (_sz, [_i64]),
    "csplat_mask_to_map": (_i, [_vp, _i64, _vp, C.c_int32, _vp, _vp, _vp]),
    "csplat_rows_scatter": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "csplat_gauss_act_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_gauss_act_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_project_points": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "csplat_image_loss_scratch_bytes": (_sz, [_i64, _i, _i, _i]),
    "csplat_image_loss_fwd": (_i, [_vp, _i64, _i, _i, _i, C.POINTER(C.c_float), _vp, _vp, _vp, _i, _f, _f, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_image_loss_bwd": (_i, [_vp, _i64, _i, _i, _i, C.POINTER(C.c_float), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "csplat_step_stats": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp]),
    "csplat_psnr_scratch_bytes": (_sz, [_i64]),
    "csplat_psnr": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "csplat_sim_hidden_fwd": (_i, [_vp, _i, _i] + [_vp] * 7),
    "csplat_sim_hidden_bwd": (_i, [_vp, _i, _i] + [_vp] * 10),
    "csplat_sim_hidden_scratch_bytes": (_sz, [_i]),
    "csplat_rows_dot_scratch_bytes": (_sz, [_i]),
    "csplat_cloth_regs_scratch_bytes": (_sz, [_i, _i, _i64]),
    "csplat_cloth_regs": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_rows_dot_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "csplat_rows_dot_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_l1": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "csplat_l1_masked": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "csplat_l1_signs": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "csplat_l1_signs_bwd": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _i, _vp, _vp]),
    "csplat_ssim_fwd_masked": (_i, [_vp, _i64, _i, _i, _i, C.POINTER(C.c_float), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "csplat_prof_enable": (_i, [C.c_uint]),
    "csplat_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "csplat_gnn_csr_temp_bytes": (_sz, [_i, _i64]),
    "csplat_gnn_build_csr": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "csplat_gnn_edge_combine_fwd": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "csplat_gnn_edge_combine_bwd": (_i, [_vp, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_gnn_segment_sum": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "csplat_gnn_gather_rows": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "csplat_gnn_gather_rows_absmax": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "csplat_gnn_edge_features": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "csplat_gnn_rows_chain_pack": (_i, [_vp, _i, _vp, _vp, _vp]),
    "csplat_gnn_rows_chain": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_gnn_edge_features_ordered": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "csplat_rollout_head": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_rollout_decode": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_rollout_integrate": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "csplat_gnn_edge_length_refine": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_double, C.c_double, C.c_double,
                                           C.c_double, _vp]),
    "csplat_dw128_workspace_bytes": (_sz, [_i64]),
    "csplat_dw128": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "csplat_dw128_bias": (_i, [_vp, _i64, _vp, _vp, _i, _vp, _vp, _vp]),
    "csplat_ln128_partial_floats": (_sz, [_i64]),
    "csplat_ln128_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _f, _vp, _vp]),
    "csplat_ln128_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "csplat_relu_mask_bias128": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "csplat_linear128_mode": (_i, [C.c_uint]),
    "csplat_linear128_mode_query": (C.c_uint, []),
    "csplat_gnn_node_update": (_i, [_vp, _i64] + [_vp] * 11 + [_f] + [_vp] * 5),
    "csplat_linear_narrow128": (_i, [_vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "csplat_gnn_node_update_image_bytes": (_sz, []),
    "csplat_gnn_node_update_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csplat_gnn_node_update_packed": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "csplat_gnn_edge_mlp3_image_bytes": (_sz, []),
    "csplat_gnn_edge_mlp3_pack": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "csplat_gnn_edge_mlp3": (_i, [_vp, _i64, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "csplat_gnn_edge_mlp3_mode": (_i, [_i]),
    "csplat_gnn_mlp3_rows": (_i, [_vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "csplat_absmax": (_i, [_vp, _i64, _vp, _vp]),
    "csplat_linear128": (_i, [_vp, _i64, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "csplat_linear128_ex": (_i, [_vp, _i64, _vp, _vp, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
}
for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so does not export what include/csplat.h declares
    _fn.restype = _res
    _fn.argtypes = _args

if lib.csplat_abi_version() != ABI_VERSION:
    raise ImportError(f"libcsplat.so ABI {lib.csplat_abi_version()} != expected {ABI_VERSION}; rebuild")


class CsplatError(RuntimeError):
    pass


# ---- dispatch visibility: every product function that can take a composed-torch branch on a GPU tensor
# reports it here.  FALLBACK_COUNTS[(site, kind)] counts them; with CSPLAT_STRICT=1 (or native.STRICT = True; the `-m gpu` tests
# run that way) the branch raises instead, unless its kind is inside an `allow_fallbacks(...)` block.  kind: "dtype" (not fp32),
# "layout" (stride / contiguity), "shape" (a width or form no HIP kernel of this library covers, e.g. latent size != 128),
# "mode" (an option the fused kernels do not implement).  Size thresholds below which the library GEMM is the faster GPU path are
# tuned dispatch, not fallbacks, and are not reported.
import collections as _collections
import contextlib as _contextlib

STRICT = os.environ.get("CSPLAT_STRICT", "0") not in ("", "0")
FALLBACK_COUNTS = _collections.Counter()
_ALLOWED = _collections.Counter()


@_contextlib.contextmanager
def allow_fallbacks(*kinds):
    """inside the block the given kinds of composed-torch branches ("shape", "dtype", "layout", "mode"; none given = all) are
    counted but do not raise under STRICT -- for callers that knowingly run a form the HIP kernels do not cover"""
    kinds = kinds or ("shape", "dtype", "layout", "mode")
    for k in kinds:
        _ALLOWED[k] += 1
    try:
        yield FALLBACK_COUNTS
    finally:
        for k in kinds:
            _ALLOWED[k] -= 1


def composed_fallback(site, kind, *tensors):
    """a product function is about to take its composed-torch branch for `tensors`.  CPU tensors (host-logic tests) are not the
    product path and are not reported."""
    if not any(torch.is_tensor(t) and t.is_cuda for t in tensors):
        return
    FALLBACK_COUNTS[(site, kind)] += 1
    if STRICT and _ALLOWED[kind] <= 0:
        raise CsplatError(f"csplat (strict): {site} would leave the HIP path ({kind}); pass fp32 contiguous tensors of a supported "
                          "shape, or wrap the call in csplat.native.allow_fallbacks()")


def why_not_f32c(*tensors):
    """None when every given GPU tensor is fp32 and contiguous, else the kind of the first obstacle"""
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            if t.dtype != torch.float32:
                return "dtype"
            if not t.is_contiguous():
                return "layout"
    return None


# caches of scratch buffers that are "zeroed once: the kernel leaves its ticket word at zero" (csplat_l1, the image loss, the cloth
# regularisers, csplat_sim_hidden_bwd ...) register here.  An entry point that FAILS may have left a ticket anywhere -- every later launch
# would then take the wrong "last workgroup" branch and return wrong sums silently -- so any error drops every cached buffer:
# the next call allocates and zeroes a fresh one.
TICKET_CACHES = []
# Recorded hipGraphs (csplat.train.CapturedStep, bench.py's GraphedSteps) hold RAW POINTERS into these buffers: whenever a cache is
# emptied the freed memory may be handed to another tensor while the graphs stay replayable.  Every eviction therefore bumps
# this epoch; a holder of recordings notes the epoch it recorded under and re-records (never replays) once it has moved.
SCRATCH_EPOCH = [0]


def evict_scratch(cache=None):
    """empty one ticketed scratch cache (None: all of them) and invalidate every recording that may point into it"""
    for c in (TICKET_CACHES if cache is None else [cache]):
        c.clear()
    SCRATCH_EPOCH[0] += 1


def check(rc, what):
    if rc != 0:
        evict_scratch()
        raise CsplatError(f"{what} failed (rc={rc}): {lib.csplat_last_error().decode(errors='replace')}")


def ptr(t):
    """device pointer of a tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "csplat: non-contiguous tensor passed to the C-ABI"
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device=None):
    """the current HIP stream of `device` as an integer handle (the step is bound by the host: torch.cuda.current_stream builds a Stream
    object per call, ~2 us; the raw query is ~0.3 us)"""
    if _RAW_STREAM is not None:
        idx = device.index if isinstance(device, torch.device) else device
        return _RAW_STREAM(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


# While a step is being RECORDED (stream capture) torch's current stream is the capture stream -- but the recording will be replayed on
# the stream that was current before.  Scratch buffers that are "zeroed once per stream" (their kernels leave the ticket words at zero,
# and launches on one stream cannot overlap) are therefore keyed on scratch_stream(): the stream the work will RUN on.
REPLAY_STREAM = {}


def scratch_stream(device=None):
    idx = device.index if isinstance(device, torch.device) else device
    idx = torch.cuda.current_device() if idx is None else idx
    alias = REPLAY_STREAM.get(idx)
    return alias if alias is not None else stream_handle(idx)


# ---- gradient sinks (csplat.dist.FlatGrads): where the LAST kernel that forms a parameter's gradient should write it.  A
# view-parallel step all-reduces ONE flat buffer; with a sink registered for a parameter, the node that finishes its gradient (K8, the
# activation / mesh-transform adjoints, the simulator's backward) writes straight into the parameter's slice of that buffer and returns a
# fresh view of it, which autograd adopts as `.grad` (no other owner: AccumulateGrad steals it) -- instead of writing a temporary that
# an in-place add then folds into a zero-filled slice (a second read-modify-write of every gradient byte, one stock launch per tensor).
GRAD_SINK = {}


def grad_sink(t):
    """the registered sink of parameter `t` as a KEY (flat buffer, offset, shape, id), or None.  Record it in forward(); grad_out() turns
    it into a tensor inside backward() -- a view created earlier and kept on the ctx would have a second owner, and AccumulateGrad only
    adopts a gradient tensor nobody else holds (it clones otherwise)"""
    e = GRAD_SINK.get(id(t))
    if e is None or e[0]() is not t:
        return None
    return e[1], e[2], e[3], id(t)


def grad_out(sink, shape, device):
    """the gradient buffer a backward writes: a FRESH view of the sink when there is one (and it fits), else fresh memory.
    A sink serves ONE writer per bind(): a second node that finishes a gradient of the same parameter inside the same step (per-view
    transform nodes, a caller's own use between bind and unbind) gets fresh memory -- two aliases of one slice would be summed by
    autograd as 2x the second contribution, silently; the post-accumulate hook then copies the sum into the slice."""
    if sink is not None:
        flat, off, shp = sink[:3]
        ent = GRAD_SINK.get(sink[3]) if len(sink) > 3 else None
        if ent is not None and ent[1] is flat and not ent[4][0] and tuple(shp) == tuple(shape) and flat.dtype == torch.float32 and \
                flat.device == device:
            n = 1
            for d in shp:
                n *= d
            v = flat[off:off + n].view(tuple(shp))
            if v.data_ptr() % 16 == 0:
                ent[4][0] = True
                return v
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def grad_release(sinks):
    """give back sinks whose views were never handed to autograd (a backward plan that is rebuilt before use)"""
    for sk in sinks:
        ent = GRAD_SINK.get(sk[3]) if (sk is not None and len(sk) > 3) else None
        if ent is not None and ent[1] is sk[0]:
            ent[4][0] = False


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """`with on_device(device)` that costs nothing when `device` is already the current one (the usual case)"""
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CsplatError("csplat: expected a GPU (HIP) tensor; this path has no CPU fallback")


class CsplatView(C.Structure):
    """mirror of `csplat_view` (include/csplat.h)"""
    _fields_ = ([("stream", _vp)] + [(n, _i) for n in ("P", "D", "M", "W", "H", "prefiltered")] +
                [(n, _f) for n in ("scale_modifier", "tanfovx", "tanfovy")] +
                [(n, _vp) for n in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                    "cov3D_precomp", "view", "proj", "campos", "alloc_ctx", "out_color", "out_depth", "radii")] +
                [("num_rendered", _i), ("layout_rendered", _i)] + [(n, _vp) for n in ("geom", "binning", "image", "dL_dpix", "scratch")] +
                [("accmask", C.c_uint)] +
                [(n, _vp) for n in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D",
                                    "dL_dsh", "dL_dscale", "dL_drot")] + [("busy_tiles", _i), ("valid", _vp)])


ACC_OPACITY, ACC_COLOR, ACC_MEAN3D, ACC_COV3D, ACC_SH, ACC_SCALE, ACC_ROT = 1, 2, 4, 8, 16, 32, 64


class ChunkAllocator:
    """Answers csplat_alloc_fn with torch-owned byte buffers and keeps them alive."""

    def __init__(self, device):
        self.device = device
        self.chunks = {}
        self.cb = ALLOC_FN(self._alloc)

    def _alloc(self, _ctx, chunk, nbytes):
        try:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self.chunks[int(chunk)] = buf
            return buf.data_ptr()
        except Exception:  # never let an exception cross the C boundary
            return None


def group_by_key(keys, n_keys):
    """(rowptr int32 [n_keys+1], perm int32 [len(keys)]): element ids grouped by key, ascending inside a group -- the stable
    argsort of `keys` as a counting sort on the device (csplat_gnn_build_csr); keys int64 in [0, n_keys)."""
    require_cuda(keys)
    assert keys.dtype == torch.int64 and keys.dim() == 1
    keys = keys.contiguous()
    E, dev = int(keys.shape[0]), keys.device
    rowptr = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    tmp = torch.empty(max(int(lib.csplat_gnn_csr_temp_bytes(n_keys, E)), 256), dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(lib.csplat_gnn_build_csr(stream_handle(dev), n_keys, E, ptr(keys), ptr(rowptr), ptr(perm), ptr(tmp)), "csplat_gnn_build_csr")
    return rowptr, perm[:E]


PROF_CLASSES = ["K1_preprocess", "K2_scan", "K3_emit_keys", "K4_radix_sort", "K5_tile_ranges", "K6_render_fwd",
                "K7_render_bwd", "K8_preprocess_bwd", "K9_dist2", "GNN"]


def prof_enable(classes=()):
    mask = 0
    for c in classes:
        mask |= 1 << (PROF_CLASSES.index(c) if isinstance(c, str) else int(c))
    check(lib.csplat_prof_enable(mask), "csplat_prof_enable")


def prof_read(cls):
    k = PROF_CLASSES.index(cls) if isinstance(cls, str) else int(cls)
    ms, n = C.c_double(0), _i64(0)
    check(lib.csplat_prof_read(k, C.byref(ms), C.byref(n)), "csplat_prof_read")
    return ms.value, n.value


# experiment hook: CSPLAT_DEBUG_FLAGS=<int> applies csplat_debug_flags at import (A/B runs of bench.py without code edits)
if os.environ.get("CSPLAT_DEBUG_FLAGS"):
    check(lib.csplat_debug_flags(int(os.environ["CSPLAT_DEBUG_FLAGS"], 0)), "csplat_debug_flags")
