"""oracle/raster_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front end of oracle/raster_ref.c and oracle/knn_ref.c (the CPU restatement of the
rasterizer the reference calls at gaussian_renderer/__init__.py:156 and of distCUDA2,
scene_reconstruction/gaussian_mesh.py:250).  PARITY UNPINNED -- see raster_ref.c header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f"liboracle_f{b}.so")) for b in (32, 64))
    if not need:
        src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("raster_ref.c", "knn_ref.c", "Makefile"))
        so_m = min(os.path.getmtime(os.path.join(_HERE, f"liboracle_f{b}.so")) for b in (32, 64))
        need = src_m > so_m
    if need:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])


def lib(dtype=np.float32):
    bits = 32 if np.dtype(dtype) == np.float32 else 64
    if bits not in _LIBS:
        path = os.path.join(_HERE, f"liboracle_f{bits}.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_preprocess.restype = C.c_int64
        L.oracle_bin.restype = C.c_int
        L.oracle_num_threads.restype = C.c_int
        assert L.oracle_real_bytes() == bits // 8
        _LIBS[bits] = L
    return _LIBS[bits]


def num_threads():
    return lib(np.float32).oracle_num_threads()


def set_threads(n):
    """OpenMP threads of both builds (small problems run faster on a few threads than on all of a 128-thread host)"""
    for dt in (np.float32, np.float64):
        lib(dt).oracle_set_threads(C.c_int(int(n)))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def forward(means3D, opacities, view, proj, campos, tanfovx, tanfovy, W, H, bg, shs=None, sh_degree=0,
            colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_mod=1.0,
            dtype=np.float32, stages="all"):
    """Run K1..K6 on the CPU.  Returns a namespace holding every intermediate buffer.

    view / proj are the reference's transposed matrices (cameras.py:63-67), any shape with 16 elements.
    """
    rt = np.dtype(dtype)
    L = lib(rt)
    real = C.c_float if rt == np.float32 else C.c_double
    means3D = _c(means3D, rt); P = means3D.shape[0]
    opacities = _c(opacities, rt).reshape(-1)
    shs = _c(shs, rt); colors_precomp = _c(colors_precomp, rt)
    scales = _c(scales, rt); rotations = _c(rotations, rt); cov3D_precomp = _c(cov3D_precomp, rt)
    view = _c(view, rt).reshape(-1); proj = _c(proj, rt).reshape(-1); campos = _c(campos, rt).reshape(-1)
    bg = _c(bg, rt).reshape(-1)
    assert (shs is None) != (colors_precomp is None)
    assert (cov3D_precomp is None) != (scales is None or rotations is None)
    M = 0 if shs is None else shs.shape[1]
    o = SimpleNamespace(P=P, W=W, H=H, M=M, D=sh_degree, dtype=rt)
    o.depth = np.zeros(P, rt); o.radii = np.zeros(P, np.int32); o.xy = np.zeros((P, 2), rt)
    o.conic_opacity = np.zeros((P, 4), rt); o.rgb = np.zeros((P, 3), rt); o.clamped = np.zeros((P, 3), np.uint8)
    o.cov3D = np.zeros((P, 6), rt); o.tiles_touched = np.zeros(P, np.uint32); o.rect = np.zeros((P, 4), np.int32)
    R = L.oracle_preprocess(C.c_int(P), C.c_int(sh_degree), C.c_int(M), C.c_int(W), C.c_int(H), _p(means3D), _p(shs),
                            _p(colors_precomp), _p(opacities), _p(scales), real(scale_mod), _p(rotations),
                            _p(cov3D_precomp), _p(view), _p(proj), _p(campos), real(tanfovx), real(tanfovy),
                            _p(o.depth), _p(o.radii), _p(o.xy), _p(o.conic_opacity), _p(o.rgb), _p(o.clamped),
                            _p(o.cov3D), _p(o.tiles_touched), _p(o.rect))
    o.R = int(R)
    if stages == "preprocess":
        return o
    gx, gy = (W + 15) // 16, (H + 15) // 16
    n = max(o.R, 1)
    o.keys_unsorted = np.zeros(n, np.uint64); o.keys = np.zeros(n, np.uint64); o.ids = np.zeros(n, np.uint32)
    o.ranges = np.zeros((gx * gy, 2), np.int32)
    rc = L.oracle_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(o.depth), _p(o.radii), _p(o.xy), _p(o.tiles_touched),
                      C.c_int64(o.R), _p(o.keys_unsorted), _p(o.keys), _p(o.ids), _p(o.ranges))
    assert rc == 0, rc
    o.keys_unsorted = o.keys_unsorted[:o.R]; o.keys = o.keys[:o.R]; o.ids = o.ids[:o.R]
    if stages == "bin":
        return o
    o.color = np.zeros((3, H, W), rt); o.out_depth = np.zeros((1, H, W), rt)
    o.final_T = np.zeros((H, W), rt); o.n_contrib = np.zeros((H, W), np.uint32)
    ids = o.ids if o.R > 0 else np.zeros(1, np.uint32)
    L.oracle_render_fwd(C.c_int(W), C.c_int(H), _p(o.ranges), _p(ids), _p(o.xy), _p(o.conic_opacity), _p(o.rgb),
                        _p(o.depth), _p(bg), _p(o.color), _p(o.out_depth), _p(o.final_T), _p(o.n_contrib))
    o._inputs = SimpleNamespace(means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                                scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, view=view, proj=proj,
                                campos=campos, bg=bg, tanfovx=tanfovx, tanfovy=tanfovy, scale_mod=scale_mod)
    return o


def backward(o, dL_dcolor_img):
    """Run K7 + K8 on the CPU for the forward state `o` (from forward())."""
    rt = o.dtype
    L = lib(rt)
    real = C.c_float if rt == np.float32 else C.c_double
    i = o._inputs
    P, W, H, M = o.P, o.W, o.H, o.M
    dpix = _c(dL_dcolor_img, rt)
    g = SimpleNamespace()
    g.mean2D = np.zeros((P, 3), rt); g.conic = np.zeros((P, 4), rt); g.opacity = np.zeros(P, rt)
    g.color = np.zeros((P, 3), rt)
    ids = o.ids if o.R > 0 else np.zeros(1, np.uint32)
    L.oracle_render_bwd(C.c_int(P), C.c_int(W), C.c_int(H), _p(o.ranges), _p(ids), _p(o.xy), _p(o.conic_opacity),
                        _p(o.rgb), _p(i.bg), _p(o.final_T), _p(o.n_contrib), _p(dpix), _p(g.mean2D), _p(g.conic),
                        _p(g.opacity), _p(g.color))
    g.mean3D = np.zeros((P, 3), rt); g.cov3D = np.zeros((P, 6), rt)
    g.sh = None if i.shs is None else np.zeros((P, M, 3), rt)
    use_pre = i.cov3D_precomp is not None
    g.scale = None if use_pre else np.zeros((P, 3), rt)
    g.rot = None if use_pre else np.zeros((P, 4), rt)
    L.oracle_preprocess_bwd(C.c_int(P), C.c_int(o.D), C.c_int(M), C.c_int(W), C.c_int(H), _p(i.means3D), _p(i.shs),
                            _p(o.clamped), _p(i.scales), real(i.scale_mod), _p(i.rotations), _p(o.cov3D),
                            C.c_int(int(use_pre)), _p(i.view), _p(i.proj), _p(i.campos), real(i.tanfovx),
                            real(i.tanfovy), _p(o.radii), _p(g.mean2D), _p(g.conic), _p(g.color), _p(g.mean3D),
                            _p(g.cov3D), _p(g.sh), _p(g.scale), _p(g.rot))
    return g


def dist2(points, dtype=np.float32):
    """Brute-force mean squared distance to the 3 nearest other points (distCUDA2 restatement)."""
    rt = np.dtype(dtype)
    pts = _c(points, rt)
    out = np.zeros(pts.shape[0], rt)
    lib(rt).oracle_dist2(C.c_int(pts.shape[0]), _p(pts), _p(out))
    return out
