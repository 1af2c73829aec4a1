"""Parity of the HIP rasterizer (through the C-ABI, via the diff_gaussian_rasterization drop-in) against the
oracle.  Bars (BASELINE.json north_star / BASELINE.md): tile & bin indices bit-exact; RGB, depth and gradients
within 1e-4 relative in fp32 (tolerance written at each assert)."""
import numpy as np
import pytest

import util
from util import image_err, make_case, oracle_forward, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-4  # north_star: "rendered RGB/depth and gradients within 1e-4 rel fp32"


def _run_gpu(case, dpix, scale_mod=1.0, **over):
    import diff_gaussian_rasterization as dgr
    inp = util.gpu_inputs(case)
    rs = util.gpu_settings(case, scale_mod=scale_mod, sh_degree=over.pop("sh_degree", None))
    kw = dict(shs=inp["shs"], colors_precomp=None, scales=inp["scales"], rotations=inp["rotations"], cov3D_precomp=None)
    for k, v in over.items():
        kw[k] = None if v is None else torch.tensor(np.asarray(v, np.float32), device="cuda", requires_grad=True)
    color, radii, depth = dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"],
                                                     opacities=inp["opacities"], **kw)
    (color * torch.tensor(dpix, device="cuda")).sum().backward()
    torch.cuda.synchronize()
    return inp, kw, color, radii, depth


CASES = [
    dict(P=2000, W=128, H=96, seed=7, grid=20, scale_mul=1.0),
    dict(P=3000, W=200, H=136, seed=8, grid=16, scale_mul=2.5),     # ragged: W,H not multiples of 16
    dict(P=800, W=64, H=64, seed=9, grid=10, scale_mul=4.0, radius=1.2),  # close camera: frustum clamp + culling
]


@pytest.mark.parametrize("path", ["tile_bucket_lds_sort", "global_radix_sort"])
@pytest.mark.parametrize("cfg", CASES)
def test_indices_bit_exact(cfg, path):
    """both binning paths (default: per-tile buckets + in-LDS bitonic sort; fallback for very long tile lists: global
    stable LSD radix sort) must reproduce the oracle's sorted (tile | depth) lists bit for bit."""
    from csplat import native
    case = make_case(**cfg)
    o = oracle_forward(case)
    try:
        native.lib.csplat_debug_flags(2 if path == "global_radix_sort" else 0)
        color, radii, depth, st = util.gpu_forward_raw(case)
    finally:
        native.lib.csplat_debug_flags(0)
    assert st["R"] == o.R
    np.testing.assert_array_equal(radii.cpu().numpy(), o.radii)
    np.testing.assert_array_equal(st["tiles_touched"], o.tiles_touched)
    if path == "global_radix_sort":
        np.testing.assert_array_equal(st["offsets"], np.cumsum(o.tiles_touched, dtype=np.uint64).astype(np.uint32))
    # per-Gaussian state that feeds the keys is bit-identical (same association order, contraction off)
    np.testing.assert_array_equal(st["depth"].view(np.uint32), o.depth.view(np.uint32))
    np.testing.assert_array_equal(st["xy"].view(np.uint32), o.xy.view(np.uint32))
    np.testing.assert_array_equal(st["conic_opacity"].view(np.uint32), o.conic_opacity.view(np.uint32))
    np.testing.assert_array_equal(st["cov3D"].view(np.uint32), o.cov3D.view(np.uint32))
    # sorted (tile|depth) keys, sorted Gaussian ids and tile ranges: bit-exact
    np.testing.assert_array_equal(st["keys"], o.keys)
    np.testing.assert_array_equal(st["ids"], o.ids)
    np.testing.assert_array_equal(st["ranges"], o.ranges)
    clamped = np.stack([(st["clamped"] >> c) & 1 for c in range(3)], 1).astype(np.uint8)
    vis = o.radii > 0
    assert (clamped[vis] != o.clamped[vis]).sum() <= 2  # SH sum can round across 0 differently only at |rgb|~1e-7
    assert rel_err(st["rgb"], o.rgb) < 1e-5


@pytest.mark.parametrize("k6", [0, 32768], ids=["k6_columns", "k6_rows"])
@pytest.mark.parametrize("cfg", CASES)
def test_forward_image(cfg, k6):
    """(k6: the default survivor-column form of the compositing forward and the row form behind csplat_debug_flags bit 15)"""
    from csplat import native
    native.lib.csplat_debug_flags(k6)
    try:
        _test_forward_image(cfg)
    finally:
        native.lib.csplat_debug_flags(0)


def _test_forward_image(cfg):
    case = make_case(**cfg)
    o = oracle_forward(case)
    o64 = oracle_forward(case, dtype=np.float64)
    color, radii, depth, st = util.gpu_forward_raw(case)
    c, d = color.cpu().numpy(), depth.cpu().numpy()
    assert image_err(c, o64.color) < TOL and image_err(d, o64.out_depth) < TOL
    assert image_err(c, o.color) < TOL and image_err(d, o.out_depth) < TOL
    assert image_err(st["final_T"], o.final_T) < TOL
    # n_contrib is an index: exact, except where an alpha / transmittance test sits within an exp() ulp of its
    # threshold (GPU v_exp_f32 vs glibc expf).  Such ties are counted and must be vanishingly rare.
    mism = (st["n_contrib"] != o.n_contrib)
    assert mism.mean() < 2e-4, f"{mism.sum()} n_contrib mismatches"


@pytest.mark.parametrize("k7", [0, 256, 32768], ids=["k7_atomics", "k7_reproducible", "k6_rows_in_front"])
@pytest.mark.parametrize("cfg", CASES)
def test_backward_grads(cfg, k7):
    """(k7: the compositing backward with float atomics (default) and in its bit-reproducible mode, csplat_debug_flags bit 8; and behind
    the ROW form of K6 (bit 15), which writes the "blended" bits K7 walks in its own way.  The batched entry point is covered by
    test_k6_forms_through_the_batched_entry_point.)"""
    from csplat import native
    native.lib.csplat_debug_flags(k7)
    try:
        _test_backward_grads(cfg)
    finally:
        native.lib.csplat_debug_flags(0)


def _test_backward_grads(cfg):
    case = make_case(**cfg)
    rng = np.random.default_rng(3)
    dpix = rng.normal(size=(3, case["H"], case["W"])).astype(np.float32)
    o64 = oracle_forward(case, dtype=np.float64)
    g64 = util.ro.backward(o64, dpix)
    inp, kw, color, radii, depth = _run_gpu(case, dpix)
    got = dict(mean3D=inp["means3D"].grad, mean2D=inp["means2D"].grad, opacity=inp["opacities"].grad.reshape(-1),
               sh=inp["shs"].grad, scale=inp["scales"].grad, rot=inp["rotations"].grad)
    for k, v in got.items():
        e = rel_err(v.cpu().numpy(), getattr(g64, k))
        assert e < TOL, (k, e)
        # per GAUSSIAN, relative to the Gaussian's own gradient (util.rowwise_rel_err; BASELINE.md "what 1e-4 rel means here"): with a
        # floor at 10 % of the tensor's largest row every Gaussian is within 1e-4; with the floor at 1 % all but a handful are (measured,
        # tools/rowwise_err_probe.py: HIP and the fp32 ORACLE itself both leave 0-3 of 800-3000 rows at 1.2-2.3e-4 against fp64 -- rows
        # whose gradient is a cancelling sum of a few hundred fp32 terms), counted and bounded
        e10 = util.rowwise_rel_err(v.cpu().numpy(), getattr(g64, k), case["P"], floor=1e-1)
        e1 = util.rowwise_rel_err(v.cpu().numpy(), getattr(g64, k), case["P"], floor=1e-2)
        assert e10.max() < TOL, (k, "floor 10 %", float(e10.max()))
        assert (e1 > TOL).sum() <= max(4, 2e-3 * case["P"]) and e1.max() < 5e-4, (k, "floor 1 %", int((e1 > TOL).sum()), float(e1.max()))


def test_colors_precomp_and_cov_precomp():
    case = make_case(P=1500, W=96, H=80, seed=11, grid=12, scale_mul=2.0)
    g = case["g"]
    rng = np.random.default_rng(5)
    colors = rng.uniform(0, 1, size=(case["P"], 3)).astype(np.float32)
    o_pre = oracle_forward(case, dtype=np.float64)  # to get cov3D
    cov = o_pre.cov3D.astype(np.float32)
    # make cov3D defined for every Gaussian (culled ones were left 0 by the oracle)
    o = oracle_forward(case, dtype=np.float64, shs=None, colors_precomp=colors, scales=None, rotations=None,
                       cov3D_precomp=cov)
    dpix = rng.normal(size=(3, case["H"], case["W"])).astype(np.float32)
    g64 = util.ro.backward(o, dpix)
    inp, kw, color, radii, depth = _run_gpu(case, dpix, shs=None, colors_precomp=colors, scales=None, rotations=None,
                                            cov3D_precomp=cov)
    assert image_err(color.detach().cpu().numpy(), o.color) < TOL
    np.testing.assert_array_equal(radii.cpu().numpy(), oracle_forward(case, shs=None, colors_precomp=colors, scales=None,
                                                                      rotations=None, cov3D_precomp=cov).radii)
    assert rel_err(kw["colors_precomp"].grad.cpu().numpy(), g64.color) < TOL
    assert rel_err(kw["cov3D_precomp"].grad.cpu().numpy(), g64.cov3D) < TOL
    assert rel_err(inp["means3D"].grad.cpu().numpy(), g64.mean3D) < TOL


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_lower_sh_degrees(deg):
    case = make_case(P=1000, W=64, H=64, seed=12, grid=10, scale_mul=3.0)
    case["sh_degree"] = deg
    o = oracle_forward(case, dtype=np.float64)
    dpix = np.random.default_rng(1).normal(size=(3, 64, 64)).astype(np.float32)
    g64 = util.ro.backward(o, dpix)
    inp, kw, color, radii, depth = _run_gpu(case, dpix)
    assert image_err(color.detach().cpu().numpy(), o.color) < TOL
    assert rel_err(inp["shs"].grad.cpu().numpy(), g64.sh) < TOL
    assert float(inp["shs"].grad[:, (deg + 1) ** 2:].abs().max()) == 0.0


def test_scale_modifier_quirk():
    """dL/dscale omits the modifier factor (upstream behaviour, raster_ref.c header)."""
    case = make_case(P=1000, W=64, H=64, seed=13, grid=10, scale_mul=2.0)
    o = oracle_forward(case, dtype=np.float64, scale_mod=1.7)
    dpix = np.random.default_rng(2).normal(size=(3, 64, 64)).astype(np.float32)
    g64 = util.ro.backward(o, dpix)
    inp, kw, color, radii, depth = _run_gpu(case, dpix, scale_mod=1.7)
    assert image_err(color.detach().cpu().numpy(), o.color) < TOL
    assert rel_err(inp["scales"].grad.cpu().numpy(), g64.scale) < TOL
    assert rel_err(inp["rotations"].grad.cpu().numpy(), g64.rot) < TOL


@pytest.mark.parametrize("form", [32768, 0], ids=["k6_rows", "k6_columns"])
def test_wave_culling_is_exact(form):
    """The ballot culling of K6/K7 may only skip entries that cannot change any pixel of the wave: image, n_contrib and
    (up to atomic order) gradients must be identical with the culling switched off (csplat_debug_flags bit 0).
    Row form of K6 (bit 15): n_contrib and final_T bit for bit -- the transmittance products run in list order and a skipped entry is a
    factor 1.0, so any entry culled although it reaches a pixel would change bits.  Survivor-column form (default): a step's sixteen
    factors are multiplied as a scan tree, and WHICH sixteen share a step depends on what was skipped -> final_T to rounding, n_contrib
    up to threshold ties (counted)."""
    from csplat import native
    case = make_case(P=3000, W=200, H=136, seed=8, grid=16, scale_mul=2.5)
    dpix = np.random.default_rng(9).normal(size=(3, case["H"], case["W"])).astype(np.float32)
    res = []
    try:
        for flag in (0, 1):
            native.lib.csplat_debug_flags(flag | form)
            color, radii, depth, st = util.gpu_forward_raw(case)
            inp, kw, c2, _, _ = _run_gpu(case, dpix)
            res.append((color.cpu().numpy(), depth.cpu().numpy(), st["n_contrib"], st["final_T"],
                        {k: inp[k].grad.cpu().numpy() for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations")}))
    finally:
        native.lib.csplat_debug_flags(0)

    def same_state(n_a, t_a, n_b, t_b):
        if form:
            np.testing.assert_array_equal(n_a, n_b)
            np.testing.assert_array_equal(t_a, t_b)
        else:
            assert float((n_a != n_b).mean()) < 2e-4
            assert rel_err(t_a, t_b) < 2e-6
    # Colour / depth: a pixel's sum is formed from partial sums whose membership depends on which entries were skipped -> equal up to
    # fp32 re-association.
    same_state(res[0][2], res[0][3], res[1][2], res[1][3])
    assert rel_err(res[0][0], res[1][0]) < 2e-6 and rel_err(res[0][1], res[1][1]) < 2e-6
    for k in res[0][4]:
        assert rel_err(res[0][4][k], res[1][4][k]) < 1e-5, k
    # the culling bound has slack: a 4x larger radius (debug bit 4) changes nothing either
    try:
        native.lib.csplat_debug_flags(16 | form)
        color, radii, depth, st = util.gpu_forward_raw(case)
    finally:
        native.lib.csplat_debug_flags(0)
    assert rel_err(color.cpu().numpy(), res[0][0]) < 2e-6
    same_state(st["n_contrib"], st["final_T"], res[0][2], res[0][3])


def test_bit_reproducible_backward_mode():
    """csplat_debug_flags bit 8: K7 keeps one record per (list entry, quadrant) and every Gaussian sums its records in emission
    order (k_det_reduce) instead of meeting in float atomics -- two runs give the same bits in EVERY gradient, and the values
    agree with the default (atomic) mode to rounding.  (the deterministic mode for tests.)"""
    from csplat import native
    case = make_case(P=3000, W=200, H=136, seed=8, grid=16, scale_mul=2.5)
    dpix = np.random.default_rng(21).normal(size=(3, case["H"], case["W"])).astype(np.float32)
    names = ("means3D", "means2D", "opacities", "shs", "scales", "rotations")
    runs = []
    try:
        for flag in (256, 256, 0):
            native.lib.csplat_debug_flags(flag)
            inp, kw, color, _, _ = _run_gpu(case, dpix)
            runs.append({k: inp[k].grad.cpu().numpy() for k in names})
    finally:
        native.lib.csplat_debug_flags(0)
    for k in names:
        np.testing.assert_array_equal(runs[0][k], runs[1][k], err_msg=k)
        assert rel_err(runs[0][k], runs[2][k]) < 1e-5, k
    g64 = util.ro.backward(oracle_forward(case, dtype=np.float64), dpix)
    assert rel_err(runs[0]["means3D"], g64.mean3D) < TOL and rel_err(runs[0]["shs"], g64.sh) < TOL


def test_empty_and_all_culled():
    import diff_gaussian_rasterization as dgr
    case = make_case(P=64, W=48, H=32, seed=14, grid=6)
    rs = util.gpu_settings(case)
    # all Gaussians behind the camera -> background image, zero radii, zero grads
    inp = util.gpu_inputs(case)
    with torch.no_grad():
        inp["means3D"] += torch.tensor(case["cam"]["camera_center"], device="cuda") * 3.0
    color, radii, depth = dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"],
                                                     shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
    color.sum().backward()
    assert int(radii.max()) == 0
    assert torch.allclose(color, torch.ones_like(color))
    assert float(depth.abs().max()) == 0.0
    assert float(inp["means3D"].grad.abs().max()) == 0.0 and float(inp["shs"].grad.abs().max()) == 0.0
    # P == 0
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    color, radii, depth = dgr.GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3),
                                                     scales=z(0, 3), rotations=z(0, 4))
    assert torch.allclose(color, torch.ones_like(color)) and radii.numel() == 0


def test_argument_errors():
    import diff_gaussian_rasterization as dgr
    case = make_case(P=16, W=32, H=32, seed=15, grid=4)
    rs = util.gpu_settings(case)
    inp = util.gpu_inputs(case)
    with pytest.raises(Exception):
        dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"],
                                   scales=inp["scales"], rotations=inp["rotations"])  # neither shs nor colours
    with pytest.raises(Exception):
        dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"],
                                   shs=inp["shs"])  # no covariance source


def test_known_answer_single_gaussian():
    """One isotropic Gaussian straight ahead: closed-form centre pixel, radius and alpha."""
    import diff_gaussian_rasterization as dgr
    from csplat import synthetic as syn
    W = H = 64
    cam = syn.make_camera(0.0, W, H, radius=4.0)
    case = dict(cam=cam, W=W, H=H, P=1, bg=np.zeros(3, np.float32), sh_degree=0)
    rs = util.gpu_settings(case)
    s = 0.05
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device="cuda")  # noqa: E731
    color, radii, depth = dgr.GaussianRasterizer(rs)(
        means3D=t([[0, 0, 0]]), means2D=t([[0, 0, 0]]), opacities=t([[0.8]]), colors_precomp=t([[1.0, 0.5, 0.25]]),
        scales=t([[s, s, s]]), rotations=t([[1, 0, 0, 0]]))
    focal = W / (2 * cam["tanfovx"])
    sigma2 = (s * focal / 4.0) ** 2 + 0.3
    assert int(radii[0]) == int(np.ceil(3 * np.sqrt(sigma2 + np.sqrt(0.1))))  # lambda = mid + sqrt(max(0.1, mid^2 - det))
    # projected centre is (W-1)/2: the 4 central pixels are at distance sqrt(0.5)
    alpha = 0.8 * np.exp(-0.5 * 0.5 / sigma2)
    c = color[:, H // 2, W // 2].cpu().numpy()
    np.testing.assert_allclose(c, alpha * np.array([1.0, 0.5, 0.25]), rtol=2e-4)
    np.testing.assert_allclose(float(depth[0, H // 2, W // 2]), alpha * 4.0, rtol=2e-4)


def test_full_size_properties():
    """BASELINE config 2 size (P=100k, 800x800): size-independent properties of the binning and the backward."""
    import diff_gaussian_rasterization as dgr
    from csplat import synthetic as syn
    sc = syn.scene_1(P=100_000, W=800, H=800, n_cams=1)
    case = dict(g=syn.gaussians_at(sc), cam=sc["cameras"][0], W=800, H=800, P=100_000, bg=sc["bg"], sh_degree=3)
    color, radii, depth, st = util.gpu_forward_raw(case)
    keys, ids, ranges = st["keys"], st["ids"], st["ranges"]
    R = st["R"]
    assert R == int(st["tiles_touched"].astype(np.int64).sum())
    assert np.all(keys[1:] >= keys[:-1])                                    # sortedness
    assert np.array_equal(np.bincount(ids, minlength=100_000), st["tiles_touched"])  # permutation of the emitted instances
    assert int((ranges[:, 1] - ranges[:, 0]).sum()) == R                      # ranges tile the list
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    nz = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
    assert np.array_equal(np.unique(tiles), nz)
    # stability: equal keys keep ascending Gaussian id
    eq = keys[1:] == keys[:-1]
    assert np.all(ids[1:][eq] > ids[:-1][eq])
    assert np.isfinite(color.cpu().numpy()).all()
    # backward is linear in dL/dpix
    rng = np.random.default_rng(0)
    dp = torch.tensor(rng.normal(size=(3, 800, 800)).astype(np.float32), device="cuda")
    grads = []
    for scale in (1.0, 2.0):
        inp = util.gpu_inputs(case)
        rs = util.gpu_settings(case)
        c, _, _ = dgr.GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"],
                                             shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
        (c * dp * scale).sum().backward()
        grads.append([inp[k].grad.clone() for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations")])
    for a, b in zip(*grads):
        assert torch.isfinite(a).all()
        assert float((2 * a - b).abs().max()) <= 2e-3 * float(b.abs().max())  # fp32 atomics reorder sums


def test_batched_views_equal_single_view_calls():
    """rasterize_views (one HIP stream per view, every csplat_forward_begin before the first csplat_forward_finish) is
    the same computation as one GaussianRasterizer call per view: images, radii, depth identical bit for bit;
    gradients of shared parameters equal up to the float-atomic accumulation order inside K7 (tolerance 1e-5 rel)."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_views
    cases = [util.make_case(P=3000, W=160, H=112, seed=5, theta=th, scale_mul=2.0) for th in (0.0, 40.0, -75.0)]
    settings = [util.gpu_settings(c) for c in cases]
    inp = util.gpu_inputs(cases[0])
    tgt = [torch.rand(3, c["H"], c["W"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(i))
           for i, c in enumerate(cases)]
    names = ("means3D", "opacities", "shs", "scales", "rotations")

    def run(batched):
        for k in names:
            inp[k].grad = None
        m2d = [torch.zeros(cases[0]["P"], 3, device="cuda", requires_grad=True) for _ in cases]
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"],
                    scales=inp["scales"], rotations=inp["rotations"]) for i in range(len(cases))]
        outs = rasterize_views(settings, kws) if batched else [GaussianRasterizer(settings[i])(**kws[i]) for i in range(len(cases))]
        loss = sum(((o[0] - t) ** 2).mean() for o, t in zip(outs, tgt))
        loss.backward()
        torch.cuda.synchronize()
        return outs, [inp[k].grad.clone() for k in names] + [m.grad.clone() for m in m2d]

    o1, g1 = run(False)
    o2, g2 = run(True)
    o3, g3 = run(True)      # stream pool reuse

    def run_stacked():      # the [V,3,H,W] batch as the differentiable output
        for k in names:
            inp[k].grad = None
        m2d = [torch.zeros(cases[0]["P"], 3, device="cuda", requires_grad=True) for _ in cases]
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"],
                    scales=inp["scales"], rotations=inp["rotations"]) for i in range(len(cases))]
        colors, outs = rasterize_views(settings, kws, stacked=True)
        assert colors.shape == (len(cases), 3, cases[0]["H"], cases[0]["W"]) and colors.requires_grad
        (((colors - torch.stack(tgt)) ** 2).mean(dim=(1, 2, 3))).sum().backward()
        torch.cuda.synchronize()
        return colors, outs, [inp[k].grad.clone() for k in names] + [m.grad.clone() for m in m2d]
    cs, os_, gs_ = run_stacked()
    for i, a in enumerate(o1):
        assert torch.equal(cs[i], a[0]) and torch.equal(os_[i][1], a[1]) and torch.equal(os_[i][2], a[2])
    for x, y in zip(g1, gs_):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) < 1e-5
    for a, b, c in zip(o1, o2, o3):
        for x, y, z in zip(a, b, c):
            assert torch.equal(x, y) and torch.equal(x, z)
    for x, y, z in zip(g1, g2, g3):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) < 1e-5
        assert rel_err(z.cpu().numpy(), x.cpu().numpy()) < 1e-5
    # a view whose image is not used gets no backward work, and sharing skips it
    def run_partial(batched):
        for k in names:
            inp[k].grad = None
        m2d = [torch.zeros(cases[0]["P"], 3, device="cuda", requires_grad=True) for _ in cases]
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"],
                    scales=inp["scales"], rotations=inp["rotations"]) for i in range(len(cases))]
        outs = rasterize_views(settings, kws) if batched else [GaussianRasterizer(settings[i])(**kws[i]) for i in range(len(cases))]
        (((outs[1][0] - tgt[1]) ** 2).mean() + ((outs[2][0] - tgt[2]) ** 2).mean()).backward()
        torch.cuda.synchronize()
        assert m2d[0].grad is None
        return [inp[k].grad.clone() for k in names] + [m2d[1].grad.clone(), m2d[2].grad.clone()]
    for x, y in zip(run_partial(False), run_partial(True)):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) < 1e-5
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rasterize_views(settings[:1], [dict(means3D=inp["means3D"], means2D=None, opacities=inp["opacities"],
                                             scales=inp["scales"], rotations=inp["rotations"])])


def test_batched_views_mixed_sizes_and_empty_view():
    """rasterize_views with views of different resolution / Gaussian count in one call, one of them empty (P = 0)."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_views
    c1 = util.make_case(P=1500, W=96, H=64, seed=3, scale_mul=2.0)
    c2 = util.make_case(P=700, W=200, H=40, seed=9, theta=30.0, scale_mul=3.0)
    s1, s2 = util.gpu_settings(c1), util.gpu_settings(c2)
    i1, i2 = util.gpu_inputs(c1), util.gpu_inputs(c2)
    e = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    empty = dict(means3D=e(0, 3), means2D=e(0, 3), opacities=e(0, 1), shs=e(0, 16, 3), scales=e(0, 3), rotations=e(0, 4))
    kws = [{k: i1[k] for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations")}, empty,
           {k: i2[k] for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations")}]
    outs = rasterize_views([s1, s1, s2], kws)
    ref1, ref2 = GaussianRasterizer(s1)(**kws[0]), GaussianRasterizer(s2)(**kws[2])
    for a, b in zip(outs[0], ref1):
        assert torch.equal(a, b)
    for a, b in zip(outs[2], ref2):
        assert torch.equal(a, b)
    bgimg = s1.bg.view(3, 1, 1).expand(3, 64, 96)
    assert torch.equal(outs[1][0], bgimg) and outs[1][1].numel() == 0 and float(outs[1][2].abs().max()) == 0.0
    (outs[0][0].mean() + outs[1][0].mean() + outs[2][0].mean()).backward()
    assert i1["means3D"].grad is not None and i2["shs"].grad is not None
    assert torch.isfinite(i1["means3D"].grad).all() and torch.isfinite(i2["shs"].grad).all()


def test_two_phase_forward_tickets_and_errors():
    """csplat_forward_begin / _finish bookkeeping: an unknown ticket is an error (text via csplat_last_error), at most 64
    tickets may be open, finishing releases them, and a finished ticket cannot be finished twice."""
    import ctypes as C
    from csplat import native as n
    case = util.make_case(P=300, W=64, H=48, seed=2, scale_mul=2.0)
    inp = util.gpu_inputs(case, requires_grad=False)
    st = util.gpu_settings(case)
    P, W, H = case["P"], case["W"], case["H"]
    keep = []

    def begin():
        alloc = n.ChunkAllocator(inp["means3D"].device)
        radii = torch.empty(P, dtype=torch.int32, device="cuda")
        tk = C.c_int(-1)
        rc = n.lib.csplat_forward_begin(
            n.stream_handle(radii.device), P, 3, 16, n.ptr(st.bg), W, H, n.ptr(inp["means3D"]), n.ptr(inp["shs"]), None,
            n.ptr(inp["opacities"]), n.ptr(inp["scales"]), 1.0, n.ptr(inp["rotations"]), None, n.ptr(st.viewmatrix),
            n.ptr(st.projmatrix), n.ptr(st.campos), float(st.tanfovx), float(st.tanfovy), 0, alloc.cb, None, n.ptr(radii), C.byref(tk))
        keep.append((alloc, radii))
        return rc, tk.value

    def finish(tk):
        color = torch.empty(3, H, W, device="cuda"); depth = torch.empty(1, H, W, device="cuda")
        R = C.c_int(0); g, b, i = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = n.lib.csplat_forward_finish(tk, n.ptr(color), n.ptr(depth), C.byref(R), C.byref(g), C.byref(b), C.byref(i))
        return rc, color, R.value

    assert finish(7)[0] != 0 and b"unknown ticket" in n.lib.csplat_last_error()
    tickets = []
    for _ in range(64):
        rc, tk = begin()
        assert rc == 0
        tickets.append(tk)
    assert len(set(tickets)) == 64
    rc, _ = begin()
    assert rc != 0 and b"64" in n.lib.csplat_last_error()                     # the 65th open forward is refused
    ref = None
    for tk in tickets:
        rc, color, R = finish(tk)
        assert rc == 0 and R > 0
        ref = color if ref is None else ref
        assert torch.equal(color, ref)                                         # 64 interleaved forwards, one result
    assert finish(tickets[0])[0] != 0                                          # released
    rc, tk = begin()
    assert rc == 0 and finish(tk)[0] == 0                                      # and usable again
    torch.cuda.synchronize()


def _wild_case(seed):
    """random cloud far from scene_1's statistics: Gaussians in front of, beside and BEHIND the camera and across the near
    plane, footprints from sub-pixel to half the image, 100:1 anisotropy, opacities below the 1/255 threshold and at 0.99+,
    ragged image sizes."""
    from csplat import synthetic as syn
    rng = np.random.default_rng(seed)
    P = int(rng.integers(40, 2500))
    W, H = int(rng.integers(17, 180)), int(rng.integers(17, 150))
    cam = syn.make_camera(float(rng.uniform(-180, 180)), W, H, phi_deg=float(rng.uniform(-80, 10)),
                          radius=float(rng.uniform(0.3, 5.0)), fovx=float(rng.uniform(0.3, 1.6)))
    means = rng.uniform(-1.5, 1.5, (P, 3)) * rng.choice([0.3, 1.0, 3.0])
    # 80 %: base size over 2.5 decades, anisotropy up to 10:1; 20 %: needles / flakes, any axis anywhere in 2e-3 .. 0.6 (300:1)
    scales = np.exp(rng.uniform(np.log(2e-3), np.log(0.5), (P, 1))) * np.exp(rng.uniform(np.log(0.1), 0.0, (P, 3)))
    needle = rng.random(P) < 0.2
    scales[needle] = np.exp(rng.uniform(np.log(2e-3), np.log(0.6), (int(needle.sum()), 3)))
    quats = rng.normal(size=(P, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0, 3.0, (P, 1))))
    opac[rng.random(P) < 0.05] = 0.002            # below 1/255: never contributes
    opac[rng.random(P) < 0.05] = 0.9995
    shs = np.concatenate([rng.normal(0, 1.0, (P, 1, 3)), rng.normal(0, 0.3, (P, 15, 3))], 1)
    g = dict(means3D=means.astype(np.float32), scales=scales.astype(np.float32), rotations=quats.astype(np.float32),
             opacities=opac.astype(np.float32), shs=shs.astype(np.float32))
    return dict(g=g, cam=cam, W=W, H=H, P=P, bg=rng.random(3).astype(np.float32), sh_degree=int(rng.integers(0, 4)))


def _fuzz_seeds():
    import os
    lo, hi = (int(v) for v in os.environ.get("CSPLAT_FUZZ_SEEDS", "100:116").split(":"))   # (wider sweeps: 200:400 ...)
    return list(range(lo, hi))


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_randomized_wild_scenes_forward_and_backward(seed):
    """fuzz: indices bit-exact, image / depth / gradients within 1e-4 against the oracle on scenes unlike scene_1."""
    case = _wild_case(seed)
    o = oracle_forward(case)
    o64 = oracle_forward(case, dtype=np.float64)
    rng = np.random.default_rng(seed)
    dpix = rng.normal(size=(3, case["H"], case["W"])).astype(np.float32)
    inp, kw, color, radii, depth = _run_gpu(case, dpix, sh_degree=case["sh_degree"])
    np.testing.assert_array_equal(radii.cpu().numpy(), o.radii)
    c, d = color.detach().cpu().numpy(), depth.detach().cpu().numpy()
    assert np.isfinite(c).all() and np.isfinite(d).all()
    # (bar: 1e-4, or twice what the fp32 oracle itself achieves against the fp64 one on this scene)
    assert image_err(c, o64.color, outlier_frac=1e-3) < max(TOL, 2 * image_err(o.color, o64.color, outlier_frac=1e-3))
    assert image_err(d, o64.out_depth, outlier_frac=1e-3) < max(TOL, 2 * image_err(o.out_depth, o64.out_depth, outlier_frac=1e-3))
    g64, g32 = util.ro.backward(o64, dpix), util.ro.backward(o, dpix)
    got = dict(mean3D=inp["means3D"].grad, mean2D=inp["means2D"].grad, opacity=inp["opacities"].grad.reshape(-1),
               sh=inp["shs"].grad, scale=inp["scales"].grad, rot=inp["rotations"].grad)
    P = case["P"]
    errs = {}
    for k, v in got.items():
        a = v.cpu().numpy().astype(np.float64).reshape(P, -1)
        b = np.asarray(getattr(g64, k), np.float64).reshape(P, -1)
        c = np.asarray(getattr(g32, k), np.float64).reshape(P, -1)
        assert np.isfinite(a).all(), k
        s_ = np.abs(b).max() + 1e-30
        errs[k] = (np.abs(a - b).max(1) / s_, np.abs(c - b).max(1) / s_)      # per Gaussian, relative to the largest gradient
    # The needles of this cloud are ill-conditioned in fp32: the fp32 ORACLE misses the fp64 one by up to 0.2 on them
    # (tools/fuzz_diag.py).  The bar therefore reads: 1e-4 wherever fp32 arithmetic can deliver it (anisotropy <= 30:1,
    # footprint radius <= 64 px, the fp32 oracle itself within a quarter of the bar); elsewhere (sums of 1e4+ cancelling pixel
    # terms whose ORDER differs between the sequential oracle and the butterfly + atomics of K7, and from run to run) 30x the
    # fp32 oracle's own error or 1e-3; and a Gaussian on which the fp32 oracle is itself off by more than 1e-3 in any of its
    # gradients is rounding noise on both sides (seed 1390 of the 800-seed sweep: one 250:1 needle of 334 px radius, fp32
    # oracle 7e-3, HIP 5e-2 .. 0.26 from run to run, the same with culling switched off) -- it only has to stay bounded.
    sc = case["g"]["scales"].astype(np.float64)
    noise = np.zeros(P, bool)
    for eg, eo in errs.values():
        noise |= eo > 10.0 * TOL
    assert noise.sum() <= max(3, P // 100), int(noise.sum())
    # (such a needle also perturbs its neighbours: its own alpha -- a difference of 1e6-sized terms along its 300 px length --
    # changes with the association / contraction of the fp32 expression, so every Gaussian behind it sees a slightly different
    # transmittance; a scene that holds one is held to 1e-3 instead of 1e-4)
    bar = TOL if not noise.any() else 10.0 * TOL
    for k, (eg, eo) in errs.items():
        well = (sc.max(1) / sc.min(1) <= 30.0) & (o.radii <= 64) & (eo <= 0.25 * TOL) & ~noise
        assert well.mean() > 0.4
        assert (eg[well] < bar).all(), (k, "well-conditioned", float(eg[well].max()))
        rest = ~well & ~noise
        assert (eg[rest] <= np.maximum(10.0 * TOL, 30.0 * eo[rest])).all(), (k, "ill-conditioned", float(eg[rest].max()))
        assert (eg[noise] < 1.0).all(), (k, "noise-level")


@pytest.mark.parametrize("own_means", [False, True])
def test_one_k8_for_all_views_equals_per_view_k8(own_means):
    """csplat_backward_views runs ONE K8 over all views when they share the Gaussians (gradients of shared parameters summed in
    registers, per-view outputs written per view); csplat_debug_flags bit 7 keeps the per-view K8 launches that add into the
    shared buffers one after the other.  Same sums in the same view order: equal up to K7's atomic order (1e-6 here).
    own_means: every view has its own means3D / rotations tensors (the train step's deformed copies), the rest is shared."""
    from csplat import native
    from diff_gaussian_rasterization import rasterize_views
    cases = [util.make_case(P=2500, W=144, H=96, seed=21, theta=th, scale_mul=2.0) for th in (10.0, -50.0, 95.0, 170.0)]
    settings = [util.gpu_settings(c) for c in cases]
    inp = util.gpu_inputs(cases[0])
    V = len(cases)
    tgt = [torch.rand(3, c["H"], c["W"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(40 + i))
           for i, c in enumerate(cases)]
    res = []
    try:
        for flag in (0, 128):
            native.lib.csplat_debug_flags(flag)
            for k in inp:
                inp[k].grad = None
            means = [(inp["means3D"] + 0.001 * i).detach().requires_grad_() for i in range(V)] if own_means else [inp["means3D"]] * V
            rots = [(inp["rotations"] * (1.0 + 0.01 * i)).detach().requires_grad_() for i in range(V)] if own_means \
                else [inp["rotations"]] * V
            m2d = [torch.zeros(cases[0]["P"], 3, device="cuda", requires_grad=True) for _ in range(V)]
            kws = [dict(means3D=means[i], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                        rotations=rots[i]) for i in range(V)]
            outs = rasterize_views(settings, kws)
            sum(((o[0] - t) ** 2).mean() for o, t in zip(outs, tgt)).backward()
            torch.cuda.synchronize()
            grads = [inp[k].grad.clone() for k in ("opacities", "shs", "scales")] + [m.grad.clone() for m in m2d]
            grads += [m.grad.clone() for m in means] + [r.grad.clone() for r in rots] if own_means else \
                [inp["means3D"].grad.clone(), inp["rotations"].grad.clone()]
            res.append(grads)
    finally:
        native.lib.csplat_debug_flags(0)
    assert len(res[0]) == len(res[1])
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("copies", [2, 3, 20])
def test_equal_depth_ties_sort_by_id(copies):
    """Exact clones (what densify_and_clone leaves behind until the next optimizer step) have bit-equal depths: the tile sort
    orders on the depth bytes and then puts runs of equal depth into id order -- short runs in place, runs longer than 8 (copies
    = 20) through the full-key fallback.  Sorted keys / ids / ranges bit-exact against the oracle's stable sort; so is a tile
    whose entries ALL share one depth (a fronto-parallel sheet)."""
    case = make_case(P=400, W=96, H=80, seed=31, grid=8, scale_mul=3.0)
    g = case["g"]
    rep = lambda a: np.repeat(a, copies, axis=0)  # noqa: E731
    case["g"] = {k: rep(v) for k, v in g.items()}
    case["P"] = 400 * copies
    o = oracle_forward(case)
    color, radii, depth, st = util.gpu_forward_raw(case)
    assert st["R"] == o.R
    np.testing.assert_array_equal(st["keys"], o.keys)
    np.testing.assert_array_equal(st["ids"], o.ids)
    np.testing.assert_array_equal(st["ranges"], o.ranges)
    assert image_err(color.cpu().numpy(), o.color) < TOL
    if copies == 20:      # all depths equal: Gaussians on a plane z = const in view space
        flat = make_case(P=300, W=64, H=64, seed=5, grid=8, scale_mul=3.0)
        cam = flat["cam"]
        Vm = np.asarray(cam["world_view_transform"], np.float64).reshape(4, 4)
        # move every centre along the viewing axis onto view-space depth 4.0 exactly representable steps are not needed: the
        # oracle and the kernel compute the same fp32 depth, equal for all points only if we place them by construction
        p = flat["g"]["means3D"].astype(np.float64)
        pv = p @ Vm[:3, :3] + Vm[3, :3]
        axis = Vm[:3, 2] / np.dot(Vm[:3, 2], Vm[:3, 2])
        flat["g"]["means3D"] = (p + np.outer(4.0 - pv[:, 2], axis)).astype(np.float32)
        o2 = oracle_forward(flat)
        _, _, _, st2 = util.gpu_forward_raw(flat)
        np.testing.assert_array_equal(st2["keys"], o2.keys)
        np.testing.assert_array_equal(st2["ids"], o2.ids)


@pytest.mark.parametrize("V,flag", [(2, 0), (3, 0), (5, 0), (8, 0), (9, 0), (4, 512)])
def test_all_views_launches_vs_per_view_calls(V, flag):
    """The three ways a step's views are issued -- ONE launch per stage with blockIdx.y = view (V <= 8 like views), per-view
    launches on per-view streams (V > 8, or csplat_debug_flags bit 9) and one GaussianRasterizer call per view -- give the same
    images / radii / depth bit for bit and the same gradients up to atomic order; every view has its OWN means3D and rotations
    (as the deformed cloth of a training step has) and shares SH / opacity / scales."""
    from csplat import native
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_views
    cases = [util.make_case(P=2500, W=144, H=96, seed=5, theta=-80.0 + 21.0 * i, scale_mul=2.0) for i in range(V)]
    settings = [util.gpu_settings(c) for c in cases]
    inp = util.gpu_inputs(cases[0])
    gen = torch.Generator(device="cuda").manual_seed(3)
    means = [(inp["means3D"].detach() + 0.01 * torch.randn(2500, 3, device="cuda", generator=gen)).requires_grad_() for _ in range(V)]
    rots = [torch.nn.functional.normalize(inp["rotations"].detach() + 0.1 * torch.randn(2500, 4, device="cuda", generator=gen)).requires_grad_()
            for _ in range(V)]
    tgt = torch.rand(V, 3, 96, 144, device="cuda", generator=gen)
    shared = ("opacities", "shs", "scales")

    def run(batched):
        for k in shared:
            inp[k].grad = None
        for t in means + rots:
            t.grad = None
        m2d = [torch.zeros(2500, 3, device="cuda", requires_grad=True) for _ in range(V)]
        kws = [dict(means3D=means[i], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"], rotations=rots[i])
               for i in range(V)]
        outs = rasterize_views(settings, kws) if batched else [GaussianRasterizer(settings[i])(**kws[i]) for i in range(V)]
        sum(((o[0] - tgt[i]) ** 2).mean() for i, o in enumerate(outs)).backward()
        torch.cuda.synchronize()
        return outs, [inp[k].grad.clone() for k in shared] + [t.grad.clone() for t in means + rots + m2d]
    try:
        native.lib.csplat_debug_flags(flag)
        o_b, g_b = run(True)
    finally:
        native.lib.csplat_debug_flags(0)
    o_s, g_s = run(False)
    for a, b in zip(o_b, o_s):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for a, b in zip(g_b, g_s):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_speculative_second_phase_equals_waiting_for_the_counts():
    """csplat_forward_views sizes the second phase from the PREVIOUS call's counts and reads this call's afterwards.  A sequence of
    calls whose scenes shrink, stay, and grow by 4x (the capacities no longer fit: the kernels leave the views alone and the phase
    is repeated with exact sizes) gives, call by call, the images / radii / depth / gradients of the same sequence with the
    speculation off (csplat_debug_flags bit 10), bit for bit up to the atomic order of the gradients; num_rendered stays the exact
    count while layout_rendered (>= it) is what the backward lays the chunks out with."""
    from csplat import native
    from diff_gaussian_rasterization import rasterize_views
    V = 3
    seq = [1.0, 0.9, 0.9, 2.0, 0.4, 1.1]                    # scale multipliers: list lengths go down, stay, up 4x, down, up

    def run(flags):
        out = []
        try:
            native.lib.csplat_debug_flags(flags)
            for step, mul in enumerate(seq):
                cases = [util.make_case(P=3000, W=160, H=112, seed=9, theta=-40.0 + 25.0 * i, scale_mul=2.0 * mul) for i in range(V)]
                settings = [util.gpu_settings(c) for c in cases]
                inp = util.gpu_inputs(cases[0])
                m2d = [torch.zeros(3000, 3, device="cuda", requires_grad=True) for _ in range(V)]
                kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                            rotations=inp["rotations"]) for i in range(V)]
                outs = rasterize_views(settings, kws)
                views = outs[0][0].grad_fn.views
                counts = [(v.num_rendered, v.layout_rendered) for v in views]
                gen = torch.Generator(device="cuda").manual_seed(step)
                tgt = torch.rand(V, 3, 112, 160, device="cuda", generator=gen)
                sum(((o[0] - tgt[i]) ** 2).mean() for i, o in enumerate(outs)).backward()
                torch.cuda.synchronize()
                out.append(([tuple(t.detach().clone() for t in o) for o in outs], counts,
                            [inp[k].grad.clone() for k in ("means3D", "opacities", "shs", "scales", "rotations")] + [t.grad.clone() for t in m2d]))
        finally:
            native.lib.csplat_debug_flags(0)
        return out
    spec, plain = run(0), run(1024)
    grew = 0
    for (o_s, c_s, g_s), (o_p, c_p, g_p) in zip(spec, plain):
        for a, b in zip(o_s, o_p):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert [c[0] for c in c_s] == [c[0] for c in c_p] and all(l >= r > 0 for r, l in c_s) and all(l == r for r, l in c_p)
        grew += any(l > r for r, l in c_s)
        for a, b in zip(g_s, g_p):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    assert grew >= 2                                         # the speculation was actually taken on some calls


@pytest.mark.parametrize("flags", [0, 2048, 4096], ids=["bucket_sort", "radix_only", "bucket_sort_forced_fallback"])
@pytest.mark.parametrize("P,lo,hi", [(1500, 1, 2048), (5000, 2049, 4096), (9000, 4097, 8192)])
def test_tile_sort_list_lengths(P, lo, hi, flags):
    """the in-LDS tile sort keeps 1 .. 8 keys per lane depending on the tile's list length (up to 8192 entries): short, medium and
    near-capacity lists against the oracle's sorted lists, bit for bit (the case asserts that its longest list is in the range it is
    meant for) -- through the interpolation bucket sort (default), the LSD radix sort alone (csplat_debug_flags bit 11) and the bucket
    sort with every multi-key thread region sent to the radix fallback (bit 12)"""
    from csplat import native
    case = make_case(P=P, W=64, H=48, seed=21, grid=12, scale_mul=6.0, radius=3.0)
    o = oracle_forward(case)
    longest = int((o.ranges[:, 1] - o.ranges[:, 0]).max())
    assert lo <= longest <= hi, longest
    try:
        native.lib.csplat_debug_flags(flags)
        color, radii, depth, st = util.gpu_forward_raw(case)
    finally:
        native.lib.csplat_debug_flags(0)
    assert st["R"] == o.R
    np.testing.assert_array_equal(st["keys"], o.keys)
    np.testing.assert_array_equal(st["ids"], o.ids)
    np.testing.assert_array_equal(st["ranges"], o.ranges)
    assert image_err(color.cpu().numpy(), o.color) < TOL


def test_tile_sort_depth_outliers_and_clusters():
    """the interpolation bucket sort maps depths linearly between the tile's min and max: a far OUTLIER squeezes every other key of
    the tile into a handful of buckets (thread regions longer than the in-place limit -> the radix fallback, taken by construction
    here), two depth CLUSTERS leave most buckets empty, and clones add equal-depth runs on top.  Sorted lists bit-exact vs the oracle."""
    case = make_case(P=6000, W=64, H=48, seed=33, grid=12, scale_mul=6.0, radius=3.0)
    g = case["g"]
    cam = case["cam"]
    Vm = np.asarray(cam["world_view_transform"], np.float64).reshape(4, 4)
    axis = Vm[:3, 2] / np.dot(Vm[:3, 2], Vm[:3, 2])          # moving a centre by t * axis adds t to its view-space depth
    m = g["means3D"].astype(np.float64)
    # squeeze the sheet to a depth slab 1e-4 thick (so that ~5000 keys share a few dozen fp32 depth values apart), then send a few
    # Gaussians far behind it and a cluster halfway
    pv = m @ Vm[:3, :3] + Vm[3, :3]
    m = m + np.outer((3.0 + (pv[:, 2] - pv[:, 2].mean()) * 1e-4) - pv[:, 2], axis)
    m[:5] += 60.0 * axis                                       # outliers: depth 63
    m[5:800] += 2.0 * axis                                     # second cluster at depth 5
    g["means3D"] = m.astype(np.float32)
    g["scales"] = (g["scales"] * 0.5).astype(np.float32)
    rep = lambda a: np.concatenate([a, a[1000:1200]], 0)  # noqa: E731   (200 clones: bit-equal depths)
    case["g"] = {k: rep(v) for k, v in g.items()}
    case["P"] = 6200
    o = oracle_forward(case)
    assert int((o.ranges[:, 1] - o.ranges[:, 0]).max()) > 1500
    color, radii, depth, st = util.gpu_forward_raw(case)
    assert st["R"] == o.R
    np.testing.assert_array_equal(st["keys"], o.keys)
    np.testing.assert_array_equal(st["ids"], o.ids)
    np.testing.assert_array_equal(st["ranges"], o.ranges)


def _grad_vs_oracles(name, got, g32, g64, P, tie_frac=1e-3, tie_tol=2e-2, exp_tie_frac=2e-4):
    """Gradient parity at sizes where pixels go a thousand entries deep.  The compositing rule has two discontinuous tests
    (alpha < 1/255, T(1 - alpha) < 1e-4); a pixel where two evaluations decide one of them differently moves the gradient of the
    Gaussians on it by an O(alpha) amount -- a THRESHOLD TIE, not an error.  (1) Against the fp32 oracle (the same arithmetic): <= 1e-4
    relative, except ties between v_exp_f32 and glibc's expf -- the same events the n_contrib comparison counts, and bounded by the same
    rate (exp_tie_frac of the Gaussians).  (2) Against the fp64 oracle: <= 1e-4 except the ties of fp32 arithmetic itself:
    tools/config2_diag.py shows the fp32 ORACLE off by the same amounts on the same ~20 of 100,000 Gaussians; counted (<= tie_frac of
    P), and each must be a tie of the fp32 oracle (or of the exp) as well.  Every tie is bounded by tie_tol of the gradient scale."""
    got, a32, a64 = (np.asarray(x, np.float64).reshape(P, -1) for x in (got, g32, g64))
    scale = np.abs(a64).max() + 1e-30
    e32 = np.abs(got - a32).max(1) / scale
    t32 = e32 > TOL
    assert t32.sum() <= max(exp_tie_frac * P, 2), (name, "vs fp32 oracle", int(t32.sum()), float(e32.max()))
    d = np.abs(got - a64).max(1) / scale
    ties = d > TOL
    assert ties.sum() <= tie_frac * P, (name, int(ties.sum()))
    assert max(d.max(), e32.max()) <= tie_tol, (name, float(d.max()), float(e32.max()))
    d32 = np.abs(a32 - a64).max(1) / scale
    assert np.all((d32[ties] > 0.5 * TOL) | t32[ties]), (name, "a deviation from fp64 that neither the fp32 oracle nor an exp tie explains")


_CONFIG2 = {}


def _config2():
    """BASELINE configs[1] at full size, built once per test session: scene, per-view cases, the image gradient, and (lazily, cached) the
    fp32 / fp64 oracle forwards and backwards of every view -- both full-size tests hold the HIP path to the SAME oracle run"""
    if not _CONFIG2:
        from csplat import synthetic as syn
        P, W, H, V = 100_000, 800, 800, 4
        sc = syn.scene_1(P=P, W=W, H=H, n_cams=V)
        g = syn.gaussians_at(sc)
        cases = [dict(g=g, cam=sc["cameras"][i], W=W, H=H, P=P, bg=sc["bg"], sh_degree=3) for i in range(V)]
        dpix = np.random.default_rng(11).normal(size=(V, 3, H, W)).astype(np.float32)
        _CONFIG2.update(P=P, W=W, H=H, V=V, cases=cases, dpix=dpix, oracle={})
    return _CONFIG2


def _config2_oracle(i):
    c = _config2()
    if i not in c["oracle"]:
        o, o64 = oracle_forward(c["cases"][i]), oracle_forward(c["cases"][i], dtype=np.float64)
        c["oracle"][i] = (o, o64, util.ro.backward(o, c["dpix"][i]), util.ro.backward(o64, c["dpix"][i]))
    return c["oracle"][i]


def _check_config2_against_oracle(colors, outs, chunks, layout_R, counted_R, m2d_grads, param_grads):
    """Per view: keys / ids / ranges / radii / tiles touched bit-exact vs the fp32 oracle, n_contrib ties < 2e-4, image / depth /
    final_T <= 1e-4 vs the fp64 oracle; the six gradients (shared parameters: sums over the 4 views; means2D: per view) <= 1e-4 vs
    the fp64 oracle with _grad_vs_oracles' tie accounting."""
    c = _config2()
    P, W, H, V = c["P"], c["W"], c["H"], c["V"]
    sums = {k: 0.0 for k in ("mean3D", "opacity", "sh", "scale", "rot")}
    sums32 = dict(sums)
    for i in range(V):
        o, o64, g32, g64 = _config2_oracle(i)
        assert counted_R[i] == o.R
        st = util.gpu_chunks(chunks[i], P, W, H, layout_R[i])
        np.testing.assert_array_equal(outs[i][1].cpu().numpy(), o.radii)
        np.testing.assert_array_equal(st["tiles_touched"], o.tiles_touched)
        np.testing.assert_array_equal(st["keys"][:o.R], o.keys)
        np.testing.assert_array_equal(st["ids"][:o.R], o.ids)
        np.testing.assert_array_equal(st["ranges"], o.ranges)
        mism = st["n_contrib"] != o.n_contrib
        assert mism.mean() < 2e-4, f"view {i}: {mism.sum()} n_contrib mismatches"
        # images: <= 1e-4 vs the fp32 oracle (threshold-tie pixels: < 1e-4 of the pixels); vs the fp64 oracle the ties of fp32 arithmetic
        # itself add to them (pixels go ~1200 entries deep here): < 1e-3 of the pixels
        assert image_err(colors[i].cpu().numpy(), o.color) < TOL and image_err(colors[i].cpu().numpy(), o64.color, outlier_frac=1e-3) < TOL
        dimg = outs[i][2].detach().cpu().numpy()
        assert image_err(dimg, o.out_depth) < TOL and image_err(dimg, o64.out_depth, outlier_frac=1e-3) < TOL
        assert image_err(st["final_T"], o.final_T) < TOL and image_err(st["final_T"], o64.final_T, outlier_frac=1e-3) < TOL
        _grad_vs_oracles(f"mean2D[{i}]", m2d_grads[i].cpu().numpy(), g32.mean2D, g64.mean2D, P)
        for k in sums:
            sums[k] = sums[k] + np.asarray(getattr(g64, k), np.float64)
            sums32[k] = sums32[k] + np.asarray(getattr(g32, k), np.float64)
    got = dict(mean3D=param_grads["means3D"], opacity=param_grads["opacities"].reshape(-1), sh=param_grads["shs"],
               scale=param_grads["scales"], rot=param_grads["rotations"])
    for k, v in got.items():
        _grad_vs_oracles(k, v.cpu().numpy(), sums32[k], sums[k], P, tie_frac=4e-3)


_NAMES = ("means3D", "opacities", "shs", "scales", "rotations")


def _config2_step():
    """the step both full-size tests run: rasterize_views over the 4 cameras, the image gradient `dpix`, one backward"""
    from diff_gaussian_rasterization import rasterize_views
    c = _config2()
    P, V = c["P"], c["V"]
    settings = [util.gpu_settings(cs) for cs in c["cases"]]
    inp = util.gpu_inputs(c["cases"][0])
    dp = torch.tensor(c["dpix"], device="cuda")
    zeros = torch.zeros(V, P, 3, device="cuda")
    one = torch.ones((), device="cuda")

    def run():
        for k in _NAMES:
            inp[k].grad = None
        m2d = [zeros[i].detach().requires_grad_() for i in range(V)]
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                    rotations=inp["rotations"]) for i in range(V)]
        colors, outs = rasterize_views(settings, kws, stacked=True)
        views = colors.grad_fn.views
        (colors * dp).sum().backward(gradient=one)
        return colors.detach(), outs, views, m2d, {k: inp[k].grad for k in _NAMES}
    return run


def test_config2_full_size_vs_oracle():
    """BASELINE configs[1] at FULL size through the EAGER batched path: P = 100k, 4 cameras 800x800, `rasterize_views`
    (one launch per stage, blockIdx.y = view), called twice so that the second call takes the SPECULATIVE second phase
    (chunks laid out for the previous call's counts).  Reference call being matched: gaussian_renderer/__init__.py:156-164
    (one GaussianRasterizer call per camera, train_utils.py:259-292 sums the cameras' gradients through autograd).
    Bars: _check_config2_against_oracle."""
    run = _config2_step()
    run()                                   # first call: waits for the counts
    colors, outs, views, m2d, pg = run()        # second call: speculative layout
    torch.cuda.synchronize()
    assert any(v.layout_rendered > v.num_rendered for v in views), "the speculative phase was not taken"
    _check_config2_against_oracle(colors, outs, [v.chunks for v in views], [v.layout_rendered for v in views],
                                  [v.num_rendered for v in views], [m.grad for m in m2d], pg)


def test_config2_full_size_faith_replay_vs_oracle():
    """The path bench.py TIMES, held to the oracle: the same full-size step recorded by csplat.graphs.ReplayedSteps
    -- the object bench.py's GraphedSteps wraps: `csplat_forward_views_faith` with capacities from one eager step's counts, nothing read
    back -- and replayed three times; after the last replay the chunks of THAT recording are decoded and held to the same bars as the
    eager test: keys / ids / ranges / radii bit-exact, images <= 1e-4, the five parameter gradients and the per-view means2D gradients
    <= 1e-4 against the C oracle with the tie accounting of _grad_vs_oracles.  Between replays the outputs are poisoned, so a replay
    that skipped work cannot pass on the previous one's results.  Reference call: gaussian_renderer/__init__.py:156-164."""
    import diff_gaussian_rasterization as dgr
    from csplat.graphs import ReplayedSteps
    c = _config2()
    rs = ReplayedSteps(_config2_step(), torch.device("cuda", torch.cuda.current_device()), G=2)
    assert dgr.forward_mode_is_default()
    rs.record()
    assert dgr.forward_mode_is_default(), "a launch-mode switch leaked out of the recording"
    for k in range(3):
        colors, outs, views, m2d, pg = rs.outs[k % 2]
        with torch.no_grad():           # poison what the replay must rewrite
            colors.fill_(float("nan"))
            for t in list(pg.values()) + [m.grad for m in m2d]:
                t.fill_(float("nan"))
        got = rs.step()
        assert got[0] is colors
    rs.check()
    i = rs.last()
    assert i == 0
    colors, outs, views, m2d, pg = rs.outs[i]
    counts = rs.replay_counts(i)
    assert all(int(v.layout_rendered) == rs.caps[0] for v in views), "the recording was not laid out for the capacities given on faith"
    _check_config2_against_oracle(colors, outs, [v.chunks for v in views], [int(v.layout_rendered) for v in views],
                                  [int(cn[0]) for cn in counts], [m.grad for m in m2d], pg)


@pytest.mark.parametrize("k7", [32768], ids=["k6_rows"])
def test_k6_forms_through_the_batched_entry_point(k7):
    """rasterize_views (one launch per stage for all views) with the ROW form of K6 (csplat_debug_flags bit 15) against the default
    survivor-column form: images equal to rounding (the two forms multiply the transmittance factors in a different order), every
    gradient equal to 1e-4 of its scale -- K7 walks the "blended" bits either form of K6 leaves -- on a scene deep enough for several
    256-entry segments per tile."""
    from csplat import native
    from diff_gaussian_rasterization import rasterize_views
    V = 3
    cases = [util.make_case(P=6000, W=96, H=80, seed=13, grid=12, scale_mul=5.0, theta=-30.0 + 30.0 * i, radius=3.0) for i in range(V)]
    o = oracle_forward(cases[0])
    assert int((o.ranges[:, 1] - o.ranges[:, 0]).max()) > 3 * 256
    settings = [util.gpu_settings(c) for c in cases]
    inp = util.gpu_inputs(cases[0])
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    tgt = torch.rand(V, 3, 80, 96, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))

    def run(flags):
        native.lib.csplat_debug_flags(flags)
        try:
            for k in names:
                inp[k].grad = None
            m2d = [torch.zeros(6000, 3, device="cuda", requires_grad=True) for _ in range(V)]
            kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                        rotations=inp["rotations"]) for i in range(V)]
            colors, _ = rasterize_views(settings, kws, stacked=True)
            ((colors - tgt) ** 2).mean().backward()
            torch.cuda.synchronize()
            return colors.detach().clone(), [inp[k].grad.clone() for k in names] + [m.grad.clone() for m in m2d]
        finally:
            native.lib.csplat_debug_flags(0)
    c0, g0 = run(0)
    c1, g1 = run(k7)
    if k7 & 32768:      # the two forms of K6 multiply the transmittance factors in a different order: images equal to rounding
        assert image_err(c1.cpu().numpy(), c0.cpu().numpy()) < 1e-5
        tol = 1e-4
    else:
        assert torch.equal(c0, c1)
        tol = 1e-5
    for a, b in zip(g1, g0):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < tol


def test_strict_dispatch_rejects_non_fp32_and_strided_inputs():
    """the rasterizer wrapper used to cast / move / re-lay-out any input silently.  Upstream's binding rejects a
    non-fp32 tensor (`.data<float>()`); under STRICT (every -m gpu test) so does the drop-in, and a strided per-Gaussian tensor is
    reported as a "layout" fallback; with the fallback allowed the call computes the same image.  The camera matrices, which the
    reference builds as transposed VIEWS (scene_reconstruction/cameras.py:63-67), pass as they do upstream."""
    from csplat import native
    from diff_gaussian_rasterization import GaussianRasterizer
    case = make_case(P=800, W=64, H=48, seed=5)
    rs = util.gpu_settings(case)
    inp = util.gpu_inputs(case, requires_grad=False)
    kw = dict(means3D=inp["means3D"], means2D=None, opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
    ref = GaussianRasterizer(rs)(**kw)[0]
    with pytest.raises(native.CsplatError, match="dtype"):
        GaussianRasterizer(rs)(**dict(kw, means3D=inp["means3D"].double()))
    wide = torch.zeros(800, 6, device="cuda")
    wide[:, ::2] = inp["scales"]
    with pytest.raises(native.CsplatError, match="layout"):
        GaussianRasterizer(rs)(**dict(kw, scales=wide[:, ::2]))
    with native.allow_fallbacks("dtype", "layout"):
        assert torch.equal(GaussianRasterizer(rs)(**dict(kw, means3D=inp["means3D"].double(), scales=wide[:, ::2]))[0], ref)
    # transposed-view camera matrices, as the reference's Camera builds them: no report, same image
    rs_t = rs._replace(viewmatrix=rs.viewmatrix.t().contiguous().t(), projmatrix=rs.projmatrix.t().contiguous().t())
    assert not rs_t.viewmatrix.is_contiguous()
    assert torch.equal(GaussianRasterizer(rs_t)(**kw)[0], ref)


def test_block_words_are_the_transposed_masks_and_blended_entries_reach_their_block():
    """Round 4's two bit tables of the binning chunk against the masks they are made from (layout: binning_offsets in csplat_raster.hip):
      * bmask[chunk][block] (K5b, read by K6 on the scalar unit) == bit `block` of mask16[64 chunk + l], l = 0..63 -- the ballots of the
        wave that owns the chunk -- for every chunk that holds a list entry;
      * bbits[slot][block] (K6 -> K7: which entries of a 256-entry segment the block BLENDED) only has bits where the entry REACHES the
        block (mask16), and no bit at or behind the block's last blended entry (blk_hi)."""
    case = util.make_case(P=6000, W=160, H=128, grid=40, scale_mul=2.5)
    _, _, _, st = util.gpu_forward_raw(case)
    R, W, H = int(st["R"]), case["W"], case["H"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    SEG = 256
    a256 = lambda x: (x + 255) // 256 * 256  # noqa: E731
    n = max(R, 1)
    slots = R // SEG + tiles + 1
    off = [0, a256(n * 8)]
    off.append(off[1] + a256(n * 4)); off.append(off[2] + a256((tiles + 1) * 4 + tiles * 16 * 4)); off.append(off[3] + a256(slots * 4))
    off.append(off[4] + a256(slots * 256 * 16)); off.append(off[5] + a256((n + 1) * 2)); off.append(off[6] + a256((n + 1) * 16))
    off.append(off[7] + a256((n + 1) * 16)); off.append(off[8] + a256((n + 1) * 8)); off.append(off[9] + a256(slots * 16 * (SEG // 8)))
    raw = st["_binning_raw"].cpu().numpy()
    mask16 = raw[off[5]:off[5] + 2 * R].view(np.uint16).astype(np.uint64)
    nch = (R + 63) // 64
    bm = raw[off[10]:off[10] + nch * 16 * 8].view(np.uint64).reshape(nch, 16)
    pad = np.zeros(nch * 64, np.uint64); pad[:R] = mask16
    lanes = np.arange(64, dtype=np.uint64)
    for b in range(16):
        want = (((pad.reshape(nch, 64) >> np.uint64(b)) & np.uint64(1)) << lanes).sum(1, dtype=np.uint64)
        np.testing.assert_array_equal(bm[:, b], want, err_msg=f"block {b}")
    seg_off = raw[off[2]:off[2] + 4 * (tiles + 1)].view(np.int32)
    blk_hi = raw[off[2] + 4 * (tiles + 1):off[2] + 4 * (tiles + 1) + tiles * 64].view(np.uint32).reshape(tiles, 16)
    nslots = int(seg_off[tiles])
    bb = raw[off[9]:off[9] + nslots * 16 * 32].view(np.uint64).reshape(nslots, 16, 4)
    ranges = st["ranges"]
    checked = 0
    for t in range(tiles):
        lo, hi = int(ranges[t, 0]), int(ranges[t, 1])
        for s in range((hi - lo + SEG - 1) // SEG):
            m = np.zeros(SEG, np.uint64); cnt = min(SEG, hi - lo - s * SEG); m[:cnt] = mask16[lo + s * SEG:lo + s * SEG + cnt]
            for b in range(16):
                if s * SEG >= blk_hi[t, b]:
                    continue                               # (K7 never reads the words of a segment behind the block's last blended entry)
                bits = np.array([(int(bb[seg_off[t] + s, b, w_]) >> l) & 1 for w_ in range(4) for l in range(64)], np.uint64)
                assert not np.any(bits & (((m >> np.uint64(b)) & np.uint64(1)) ^ np.uint64(1))), (t, s, b)
                pos = np.nonzero(bits)[0]
                assert pos.size == 0 or s * SEG + int(pos.max()) < blk_hi[t, b], (t, s, b)
                checked += 1
    assert checked > 50


def test_accumulation_records_are_left_zero_by_the_backward():
    """Round 5: K7's per-Gaussian accumulation records are kept from step to step, zeroed once, and K8 clears every record it consumes
    (csplat.h CSPLAT_SCRATCH_ZEROED) -- no clearing launch per step.  Two different steps in a row (other image gradients, other
    camera: other Gaussians visible) must each give the gradients of a run that starts from freshly zeroed records, and the records
    must read all zero after every backward."""
    import diff_gaussian_rasterization as dgr
    from csplat import native
    from diff_gaussian_rasterization import rasterize_views
    V, P = 2, 5000
    cases = [util.make_case(P=P, W=96, H=80, seed=17, grid=12, scale_mul=4.0, theta=-40.0 + 80.0 * i, radius=3.0) for i in range(V + 1)]
    inp = util.gpu_inputs(cases[0])
    gen = torch.Generator(device="cuda").manual_seed(5)

    def run(case_ids, dp):
        for k in _NAMES:
            inp[k].grad = None
        m2d = [torch.zeros(P, 3, device="cuda", requires_grad=True) for _ in case_ids]
        kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                    rotations=inp["rotations"]) for i in range(len(case_ids))]
        colors, _ = rasterize_views([util.gpu_settings(cases[c]) for c in case_ids], kws, stacked=True)
        (colors * dp).sum().backward()
        torch.cuda.synchronize()
        return [inp[k].grad.clone() for k in _NAMES] + [m.grad.clone() for m in m2d]
    dpA = torch.randn(V, 3, 80, 96, device="cuda", generator=gen)
    dpB = torch.randn(V, 3, 80, 96, device="cuda", generator=gen)
    run([0, 1], dpA)
    assert dgr._ACC_SCRATCH, "the persistent records were not used"
    for buf in dgr._ACC_SCRATCH.values():
        assert int(buf.view(torch.int32).abs().max()) == 0, "a backward left accumulation records behind"
    gB = run([1, 2], dpB)                    # straight behind step A, on its records
    for buf in dgr._ACC_SCRATCH.values():
        assert int(buf.view(torch.int32).abs().max()) == 0
    native.evict_scratch(dgr._ACC_SCRATCH)   # fresh records
    gB_fresh = run([1, 2], dpB)
    for a, b in zip(gB, gB_fresh):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_gradient_sink_keeps_one_writer_when_a_second_node_rebuilds_its_plan():
    """a view-parallel step binds every parameter's slice of the flat gradient buffer as its gradient SINK
    (csplat.dist.FlatGrads.bind), one writer per bind.  Two rasterizer nodes of one step share the bound parameters; the second node's
    forward MISSES its speculation (its scene grew 4x: the backward plan is rebuilt).  The rebuilt plan used to give back a sink the
    FIRST node holds and then claim the same slice: two kernels writing aliased memory, autograd summing the alias twice.  The
    gradients must be the plain sum of the two nodes' gradients (here: what the same two calls give without any sink)."""
    from csplat.dist import FlatGrads
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import rasterize_views
    V, P = 2, 2900          # (a shape no other test of the suite uses: the library's speculation history is per process and shape)
    names = ("means3D", "opacities", "shs", "scales", "rotations")

    def run(bound):
        res = None
        base = util.make_case(P=P, W=176, H=128, seed=9, theta=-40.0, scale_mul=2.0)
        inp = util.gpu_inputs(base)
        params = [inp[k] for k in names]
        fg = FlatGrads(params) if bound else None
        try:
            for rnd in range(2):                       # round 0 leaves the capacities the second node of round 1 will miss
                for p in params:
                    p.grad = None
                if fg is not None:
                    fg.bind()
                before = dict(dgr.SPEC_STATS)
                loss = 0.0
                for node, mul in enumerate((1.0, 1.0 if rnd == 0 else 2.0)):
                    cases = [util.make_case(P=P, W=176, H=128, seed=9, theta=-40.0 + 25.0 * i, scale_mul=2.0) for i in range(V)]
                    settings = [util.gpu_settings(c, scale_mod=mul) for c in cases]
                    m2d = [torch.zeros(P, 3, device="cuda", requires_grad=True) for _ in range(V)]
                    kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                rotations=inp["rotations"]) for i in range(V)]
                    outs = rasterize_views(settings, kws)
                    gen = torch.Generator(device="cuda").manual_seed(10 * rnd + node)
                    tgt = torch.rand(V, 3, 128, 176, device="cuda", generator=gen)
                    loss = loss + (1 + node) * sum(((o[0] - tgt[i]) ** 2).mean() for i, o in enumerate(outs))
                missed = dgr.SPEC_STATS["miss"] - before["miss"]
                loss.backward()
                torch.cuda.synchronize()
                if fg is not None:
                    fg.unbind()
                res = ([p.grad.detach().clone() for p in params], missed)
        finally:
            if fg is not None:
                fg.close()
        return res
    sunk, missed = run(True)              # (first: the plain run leaves the grown scene's counts in the library's per-process history)
    plain, _m0 = run(False)
    assert missed >= 1, "the second node of the last round was meant to miss its speculation"
    for name, a, b in zip(names, sunk, plain):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, name


@pytest.mark.parametrize("G", [1, 3, 4])
def test_backward_in_parts_equals_the_whole_backward(G):
    """csplat_backward_views_parts (round 6: the view-parallel step's gradient rows leave for the other ranks slice by slice): K7 once, then
    K8 for G ranges of Gaussians -- every gradient of every input BIT-equal to the one-call backward (bit-reproducible K7 mode, so that
    K7's own sums are the same in both runs), the row ranges tile [0, P) at multiples of 32, and a slice leaves the rows of the others
    untouched until its own launch."""
    from csplat import native
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import rasterize_views
    V, P = 3, 2901
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    native.lib.csplat_debug_flags(256)
    try:
        def run(parts):
            base = util.make_case(P=P, W=144, H=112, seed=4, theta=-30.0, scale_mul=2.0)
            inp = util.gpu_inputs(base)
            cases = [util.make_case(P=P, W=144, H=112, seed=4, theta=-30.0 + 30.0 * i, scale_mul=2.0) for i in range(V)]
            settings = [util.gpu_settings(c) for c in cases]
            m2d = [torch.zeros(P, 3, device="cuda", requires_grad=True) for _ in range(V)]
            kws = [dict(means3D=inp["means3D"], means2D=m2d[i], opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                        rotations=inp["rotations"]) for i in range(V)]
            colors, _outs = rasterize_views(settings, kws, stacked=True)
            gen = torch.Generator(device="cuda").manual_seed(3)
            loss = ((colors - torch.rand(V, 3, 112, 144, device="cuda", generator=gen)) ** 2).mean()
            rows = []
            if parts:
                with dgr.deferred_k8() as h:
                    loss.backward()
                assert len(h.entries) == 1
                # (the gradients autograd holds are the buffers the slices write: poison them first -- a gradient that had been copied
                #  out before its slice ran would keep whatever the allocator's recycled memory held, e.g. the other run's values)
                for t in [inp[k].grad for k in names] + [m.grad for m in m2d]:
                    t.fill_(float("nan"))
                for g_ in range(G):
                    rows.append(h.rows(g_, G))
                    h.launch(g_, G)
                    if g_ + 1 < G:      # rows of later slices are still untouched
                        assert torch.isnan(inp["scales"].grad[rows[-1][1]:]).all() and torch.isfinite(inp["scales"].grad[:rows[-1][1]]).all()
            else:
                loss.backward()
            torch.cuda.synchronize()
            return [inp[k].grad.clone() for k in names] + [m.grad.clone() for m in m2d], rows
        whole, _ = run(False)
        cut, rows = run(True)
    finally:
        native.lib.csplat_debug_flags(0)
    assert rows[0][0] == 0 and rows[-1][1] == P and all(a[1] == b[0] for a, b in zip(rows, rows[1:])) and \
        all(lo % 32 == 0 for lo, _hi in rows)
    for a, b in zip(whole, cut):
        assert torch.equal(a, b)
