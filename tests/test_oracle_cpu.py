"""CPU suite (-m "not gpu"): the oracle against the golden vectors captured from the reference's importable modules
(tests/golden/make_golden.py), the two independent restatements against each other, host-side logic, and the C-ABI
export list.  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util
from util import golden, make_case, oracle_forward, rel_err

torch = pytest.importorskip("torch")
from csplat import synthetic as syn  # noqa: E402
from oracle import gnn_ref, raster_oracle as ro, raster_torch as rt  # noqa: E402


# ------------------------------------------------------------------ golden: camera math (cameras.py / graphics_utils.py)
def test_camera_matrices_match_reference():
    g = golden("camera.npz")
    for k in range(g["R"].shape[0]):
        R, T = syn.c2w_to_RT(g["c2w"][k])
        np.testing.assert_allclose(R, g["R"][k], atol=1e-12)
        np.testing.assert_allclose(T, g["T"][k], atol=1e-12)
        wv, full, center = syn.camera_matrices(R, T, float(g["fovx"][k]), float(g["fovy"][k]))
        np.testing.assert_allclose(wv, g["world_view_transform"][k], atol=1e-6)
        np.testing.assert_allclose(full, g["full_proj_transform"][k], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(center, g["camera_center"][k], atol=2e-5)
    np.testing.assert_allclose(syn.projection_matrix(0.01, 100.0, 0.6911, 0.5), g["proj_0p01_100"], atol=1e-7)
    np.testing.assert_allclose(syn.world_to_view(g["R"][0], g["T"][0], np.array([0.1, -0.2, 0.3]), 1.5),
                               g["w2v2_translate"], atol=1e-6)


# ------------------------------------------------------------------ golden: SH -> RGB stage of K1 (utils/sh_utils.py)
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_matches_reference_eval_sh(deg):
    g = golden("sh.npz")
    sh = np.ascontiguousarray(g["sh"].transpose(0, 2, 1))  # reference layout [N,3,16] -> rasterizer layout [N,16,3]
    dirs = g["dirs"]
    n = dirs.shape[0]
    # a camera that sees every point at depth 1 and for which normalize(p - campos) == dirs
    view = np.zeros((4, 4), np.float32); view[0, 0] = view[1, 1] = 1; view[3, 2] = 1.0; view[3, 3] = 1
    proj = np.zeros((4, 4), np.float32); proj[0, 0] = proj[1, 1] = 0.3; proj[3, 3] = 1
    for dt, tol in ((np.float32, 3e-6), (np.float64, 2e-6)):
        o = ro.forward(dirs, np.full(n, 0.5), view, proj, np.zeros(3), 0.5, 0.5, 64, 64, np.zeros(3), shs=sh,
                       sh_degree=deg, scales=np.full((n, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (n, 1)), dtype=dt,
                       stages="preprocess")
        assert (o.radii > 0).all()
        np.testing.assert_allclose(o.rgb, g[f"clamped_deg{deg}"], atol=tol)
        np.testing.assert_array_equal(o.clamped.astype(bool), (g[f"rgb_deg{deg}"] + 0.5) < 0)
    t = rt._eval_sh(deg, torch.tensor(sh, dtype=torch.float64), torch.tensor(dirs, dtype=torch.float64)).numpy()
    np.testing.assert_allclose(t, g[f"rgb_deg{deg}"], atol=2e-6)


def test_misc_golden():
    g = golden("misc.npz")
    a, b = g["psnr_a"], g["psnr_b"]
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1, keepdims=True)
    np.testing.assert_allclose(20 * np.log10(1.0 / np.sqrt(mse)), g["psnr"], rtol=1e-5)
    x = g["inv_sigmoid_in"]
    np.testing.assert_allclose(np.log(x / (1 - x)), g["inv_sigmoid"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ the two restatements agree; analytic bwd == autograd
def _small_case():
    case = make_case(P=600, W=64, H=64, seed=21, grid=12, scale_mul=3.0)
    return case


def test_c_forward_equals_torch_forward():
    case = _small_case()
    o = oracle_forward(case, dtype=np.float64)
    g = case["g"]
    T = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    color, depth, ncon = rt.render(o, T(g["means3D"]), torch.zeros(case["P"], 3, dtype=torch.float64), T(g["opacities"]),
                                   shs=T(g["shs"]), scales=T(g["scales"]), rotations=T(g["rotations"]))
    assert np.abs(color.numpy() - o.color).max() < 1e-12
    assert np.abs(depth.numpy() - o.out_depth).max() < 1e-11
    assert (ncon.numpy() != o.n_contrib).sum() == 0
    assert o.R > 500 and o.n_contrib.max() > 50  # the case exercises deep lists and early termination


@pytest.mark.parametrize("mode", ["sh_scale_rot", "precomp", "deg1_mod"])
def test_c_backward_equals_autograd(mode):
    case = _small_case()
    g = case["g"]
    P = case["P"]
    rng = np.random.default_rng(4)
    dpix = rng.normal(size=(3, case["H"], case["W"]))
    T = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)  # noqa: E731
    m3, m2, op = T(g["means3D"]), T(np.zeros((P, 3))), T(g["opacities"])
    mod = 1.0
    if mode == "precomp":
        o0 = oracle_forward(case, dtype=np.float64)
        colors = rng.uniform(0, 1, size=(P, 3))
        kw = dict(shs=None, colors_precomp=colors, scales=None, rotations=None, cov3D_precomp=o0.cov3D)
        o = oracle_forward(case, dtype=np.float64, **kw)
        col, cov = T(colors), T(o0.cov3D)
        color, _, _ = rt.render(o, m3, m2, op, colors_precomp=col, cov3D_precomp=cov)
    else:
        if mode == "deg1_mod":
            case["sh_degree"] = 1
            mod = 1.6
        o = oracle_forward(case, dtype=np.float64, scale_mod=mod)
        sh, scl, rot = T(g["shs"]), T(g["scales"]), T(g["rotations"])
        color, _, _ = rt.render(o, m3, m2, op, shs=sh, scales=scl, rotations=rot)
    gr = ro.backward(o, dpix)
    (color * torch.tensor(dpix)).sum().backward()
    # 1/(det^2 + 1e-7) guard of the analytic path vs exact autograd: <= 1e-6 relative on this case
    assert rel_err(gr.mean3D, m3.grad.numpy()) < 1e-6
    assert rel_err(gr.mean2D, m2.grad.numpy()) < 1e-12
    assert rel_err(gr.opacity, op.grad.numpy().reshape(-1)) < 1e-12
    if mode == "precomp":
        assert rel_err(gr.color, col.grad.numpy()) < 1e-12
        assert rel_err(gr.cov3D, cov.grad.numpy()) < 1e-5
    else:
        assert rel_err(gr.sh, sh.grad.numpy()) < 1e-12
        # upstream quirk: dL/dscale is the gradient w.r.t. (modifier * scale)
        assert rel_err(gr.scale * mod, scl.grad.numpy()) < 1e-5
        assert rel_err(gr.rot, rot.grad.numpy()) < 1e-5


def test_f32_oracle_tracks_f64_oracle():
    case = make_case(P=2000, W=128, H=96, seed=7, grid=20)
    o32, o64 = oracle_forward(case), oracle_forward(case, dtype=np.float64)
    assert rel_err(o32.color, o64.color) < 1e-5 and rel_err(o32.out_depth, o64.out_depth) < 1e-5
    assert (o32.radii != o64.radii).mean() < 0.01
    dpix = np.random.default_rng(0).normal(size=(3, 96, 128)).astype(np.float32)
    g32, g64 = ro.backward(o32, dpix), ro.backward(o64, dpix)
    for k in ("mean3D", "mean2D", "opacity", "sh", "scale", "rot"):
        assert rel_err(getattr(g32, k), getattr(g64, k)) < 2e-4, k


def test_binning_invariants_and_edge_cases():
    case = make_case(P=3000, W=200, H=136, seed=8, grid=16, scale_mul=2.5)  # ragged image size
    o = oracle_forward(case)
    assert o.R == int(o.tiles_touched.sum()) and o.R == len(o.keys)
    assert np.all(o.keys[1:] >= o.keys[:-1])
    eq = o.keys[1:] == o.keys[:-1]
    assert np.all(o.ids[1:][eq] > o.ids[:-1][eq])           # stable: equal (tile, depth) keeps ascending id
    assert int((o.ranges[:, 1] - o.ranges[:, 0]).sum()) == o.R
    area = (o.rect[:, 2] - o.rect[:, 0]) * (o.rect[:, 3] - o.rect[:, 1])
    np.testing.assert_array_equal(area[o.radii > 0], o.tiles_touched[o.radii > 0])
    # everything behind the camera: nothing rendered, image = background
    far = dict(case); far["g"] = dict(case["g"]); far["g"]["means3D"] = case["g"]["means3D"] + case["cam"]["camera_center"] * 3
    oz = oracle_forward(far)
    assert oz.R == 0 and int(oz.radii.max()) == 0 and np.allclose(oz.color, 1.0)


def test_binning_against_an_independent_numpy_restatement():
    """VERDICT r1 weak item 1 (independence): oracle/raster_ref.c's binning (key emission, stable sort, tile ranges) against a
    second restatement written here in numpy from the published algorithm alone -- rectangle from (centre, radius) by C-style
    truncation, instances emitted y-major, key = tile << 32 | depth bits, `np.argsort(kind="stable")`, ranges by run boundaries.
    Keys, sorted ids and ranges must be identical; so must `n_contrib` recomputed per pixel from those lists with the published
    compositing rule in float64 python (on a small image)."""
    case = make_case(P=600, W=80, H=64, seed=21, grid=10, scale_mul=3.0)
    o = oracle_forward(case)
    W, H = case["W"], case["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    keys, ids = [], []
    for i in np.nonzero(o.radii > 0)[0]:
        x, y, r = np.float32(o.xy[i, 0]), np.float32(o.xy[i, 1]), np.float32(o.radii[i])
        tr = lambda v: int(np.float32(v))                                                   # C (int) cast: truncation  # noqa: E731
        minx = min(gx, max(0, tr((x - r) / np.float32(16)))); maxx = min(gx, max(0, tr((x + r + np.float32(15)) / np.float32(16))))
        miny = min(gy, max(0, tr((y - r) / np.float32(16)))); maxy = min(gy, max(0, tr((y + r + np.float32(15)) / np.float32(16))))
        dbits = int(np.float32(o.depth[i]).view(np.uint32))
        for ty in range(miny, maxy):
            for tx in range(minx, maxx):
                keys.append(((ty * gx + tx) << 32) | dbits); ids.append(i)
    keys, ids = np.array(keys, np.uint64), np.array(ids, np.uint32)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(keys[order], o.keys)
    np.testing.assert_array_equal(ids[order], o.ids)
    tiles = (keys[order] >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((gx * gy, 2), np.int32)
    for t in np.unique(tiles):
        w = np.nonzero(tiles == t)[0]
        ranges[t] = (w[0], w[-1] + 1)
    np.testing.assert_array_equal(ranges, o.ranges)
    # n_contrib / final_T of every pixel of four tiles from the lists, published rule, python floats
    o64 = oracle_forward(case, dtype=np.float64)
    for t in np.argsort(-(ranges[:, 1] - ranges[:, 0]))[:4]:
        for py in range((t // gx) * 16, min(H, (t // gx) * 16 + 16), 3):
            for px in range((t % gx) * 16, min(W, (t % gx) * 16 + 16), 3):
                T, last = 1.0, 0
                for k in range(ranges[t, 0], ranges[t, 1]):
                    g = o64.ids[k]
                    dx, dy = o64.xy[g, 0] - px, o64.xy[g, 1] - py
                    a_, b_, c_, op = o64.conic_opacity[g]
                    power = -0.5 * (a_ * dx * dx + c_ * dy * dy) - b_ * dx * dy
                    if power > 0:
                        continue
                    alpha = min(0.99, op * np.exp(power))
                    if alpha < 1.0 / 255.0:
                        continue
                    if T * (1 - alpha) < 1e-4:
                        break
                    T *= 1 - alpha
                    last = k - ranges[t, 0] + 1
                assert last == o64.n_contrib[py, px] and abs(T - o64.final_T[py, px]) < 1e-12, (px, py)


def test_known_answer_single_gaussian_cpu():
    W = H = 64
    cam = syn.make_camera(0.0, W, H, radius=4.0)
    s = 0.05
    o = ro.forward(np.zeros((1, 3)), np.array([0.8]), cam["world_view_transform"], cam["full_proj_transform"],
                   cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H, np.zeros(3),
                   colors_precomp=np.array([[1.0, 0.5, 0.25]]), scales=np.full((1, 3), s), rotations=np.array([[1.0, 0, 0, 0]]),
                   dtype=np.float64)
    focal = W / (2 * cam["tanfovx"])
    sigma2 = (s * focal / 4.0) ** 2 + 0.3
    assert int(o.radii[0]) == int(np.ceil(3 * np.sqrt(sigma2 + np.sqrt(0.1))))  # lambda = mid + sqrt(max(0.1, mid^2 - det))
    np.testing.assert_allclose(o.xy[0], [(W - 1) / 2, (H - 1) / 2], atol=1e-4)
    alpha = 0.8 * np.exp(-0.25 / sigma2)
    np.testing.assert_allclose(o.color[:, H // 2, W // 2], alpha * np.array([1.0, 0.5, 0.25]), rtol=1e-5)
    np.testing.assert_allclose(o.out_depth[0, H // 2, W // 2], alpha * 4.0, rtol=1e-5)
    assert o.n_contrib[H // 2, W // 2] == 1 and o.n_contrib[0, 0] == 0


# ------------------------------------------------------------------ distCUDA2 restatement
def test_knn_oracle_vs_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(3000, 3)).astype(np.float32)
    pts[10] = pts[11]  # coincident pair -> one zero distance
    d64 = ro.dist2(pts, np.float64)
    dd, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.testing.assert_allclose(d64, (dd[:, 1:] ** 2).mean(1), rtol=1e-10, atol=1e-14)
    d32 = ro.dist2(pts, np.float32)
    np.testing.assert_allclose(d32, d64, rtol=1e-5)
    assert ro.dist2(pts[:4], np.float64).shape == (4,)


# ------------------------------------------------------------------ GNN restatement vs shim-derived goldens
def test_gnn_oracle_matches_reference_modules():
    g = golden("gnn.npz")
    p = {k[4:]: g[k] for k in g.files if k.startswith("epd.")}
    y = gnn_ref.encode_process_decode(p, g["epd_x"], g["edge_index"], g["epd_e"])
    assert rel_err(y, g["epd_y"]) < 1e-5
    pi = {"L." + k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith("inet.")}
    x1, e1 = gnn_ref.interaction(pi, "L", g["in_x"].astype(np.float64), g["edge_index"], g["in_e"].astype(np.float64))
    assert rel_err(x1, g["in_x_out"]) < 1e-5
    np.testing.assert_array_equal(g["in_e_out"], 2 * g["in_e"])      # SURVEY F7 in the reference's own output
    np.testing.assert_allclose(e1, g["in_e_out"], rtol=1e-7)


def test_normalizer_oracle_and_module_match_reference():
    g = golden("normalizer.npz")
    nz = gnn_ref.Normalizer(5)
    o1 = nz(g["b1"].astype(np.float64), True); o2 = nz(g["b2"].astype(np.float64), True); o3 = nz(g["b1"].astype(np.float64), False)
    for a, k in ((o1, "o1"), (o2, "o2"), (o3, "o3"), (nz.inverse(o3), "inv")):
        np.testing.assert_allclose(a, g[k], rtol=2e-4, atol=2e-5)
    from meshnet.model_utils import Normalizer
    m = Normalizer(size=5, device="cpu")
    t = lambda a: torch.tensor(a)  # noqa: E731
    r1 = m(t(g["b1"]), True); r2 = m(t(g["b2"]), True); r3 = m(t(g["b1"]), False)
    for a, k in ((r1, "o1"), (r2, "o2"), (r3, "o3"), (m.inverse(r3), "inv")):
        np.testing.assert_allclose(a.numpy(), g[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m._acc_sum.numpy(), g["acc_sum"], rtol=1e-6)
    assert float(m._acc_count) == float(g["acc_count"]) and float(m._num_accumulations) == float(g["num_acc"])


# ------------------------------------------------------------------ host logic of the drop-in modules (CPU torch)
def test_simulators_match_reference():
    g = golden("simulator.npz")
    from meshnet.meshnet_network import ResidualMeshSimulator, ResidualMeshSimulatorEmbedding, SinusoidalEncoder
    enc = SinusoidalEncoder(input_dim=1, num_freqs=6)
    assert enc.output_dim == int(g["enc_dim"]) == 13
    out = torch.stack([enc(t) for t in torch.tensor(g["enc_in"])])
    np.testing.assert_allclose(out.numpy(), g["enc_out"], atol=1e-6)
    enc3 = SinusoidalEncoder(input_dim=3, num_freqs=4, min_freq_log2=-1, scale=0.5, use_identity=False)
    np.testing.assert_allclose(enc3(torch.tensor(g["enc3_in"])).numpy(), g["enc3_out"], atol=1e-6)
    mesh = torch.tensor(g["res_mesh"])
    V = mesh.shape[1]
    sim = ResidualMeshSimulator(mesh, device="cpu")
    sd = {k[4:]: torch.tensor(g[k]) for k in g.files if k.startswith("res.")}
    assert set(sd) == set(sim.state_dict())              # checkpoint keys identical to the reference's
    sim.load_state_dict(sd)
    assert sim.time_delta == float(g["res_time_delta"])
    for tt, ref in zip(g["res_times"], g["res_out"]):
        got = sim(torch.tensor(float(tt)).repeat(V, 1))
        np.testing.assert_allclose(got.detach().numpy(), ref, atol=1e-5)
    assert ResidualMeshSimulator(mesh[:1], device="cpu").time_delta == float(g["res1_time_delta"]) == 1.0
    # forward_times: the cameras of a step at once (what render_views uses) == forward() per time, same error behaviour
    both = sim.forward_times([float(tt) for tt in g["res_times"]])
    np.testing.assert_allclose(both.detach().numpy(), g["res_out"], atol=1e-5)
    with pytest.raises(ValueError):
        sim.forward_times([0.0, 1.3])
    # the per-time-set cache follows the table: edit it in place, then replace it
    t0 = float(g["res_times"][0])
    before = sim.forward_times([t0]).detach().clone()
    with torch.no_grad():
        sim.mesh_predictions += 1.0
    np.testing.assert_allclose(sim.forward_times([t0]).detach().numpy(), before.numpy() + 1.0, atol=1e-6)
    sim.mesh_predictions = sim.mesh_predictions + 1.0                 # replaced by a new tensor
    np.testing.assert_allclose(sim.forward_times([t0]).detach().numpy(), before.numpy() + 2.0, atol=1e-6)
    with torch.no_grad():
        mesh -= 1.0                                                   # (`mesh` is the tensor that was edited in place)
    sim.mesh_predictions = mesh
    np.testing.assert_allclose(sim.forward_times([t0]).detach().numpy(), before.numpy(), atol=1e-6)
    with pytest.raises(ValueError):
        sim(torch.tensor(1.3).repeat(V, 1))
    assert int(g["res_oob_raises"]) == 1
    emb = ResidualMeshSimulatorEmbedding(mesh, device="cpu")
    emb.load_state_dict({k[4:]: torch.tensor(g[k]) for k in g.files if k.startswith("emb.")})
    for tt, ref in zip(g["res_times"], g["emb_out"]):
        np.testing.assert_allclose(emb(torch.tensor(float(tt)).repeat(V, 1)).detach().numpy(), ref, atol=1e-6)


def test_gnn_module_state_dict_keys_and_fail_loud_on_cpu():
    g = golden("gnn.npz")
    from meshnet.graph_network import EncodeProcessDecode
    from meshnet.cloth_network import ClothMeshSimulator
    from csplat.native import CsplatError
    net = EncodeProcessDecode(8, 3, 4, 32, 3, 2, 32)
    ref_keys = {k[4:] for k in g.files if k.startswith("epd.")}
    assert set(net.state_dict()) == ref_keys
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == g["epd." + k].shape
    sim = ClothMeshSimulator(3, 8, 4, 32, 2, 2, 32, 2, 2, normalize=True, device="cpu")
    assert set(sim.state_dict()) == {k[4:] for k in g.files if k.startswith("sim.")}
    # the message-passing kernels have no CPU path: CPU tensors must raise, never silently fall back
    with pytest.raises(CsplatError):
        net(torch.tensor(g["epd_x"]), torch.tensor(g["edge_index"]), torch.tensor(g["epd_e"]))


def test_rasterizer_dropin_fails_loudly_without_gpu_tensors():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    from csplat.native import CsplatError
    z = torch.zeros
    rs = GaussianRasterizationSettings(16, 16, 0.5, 0.5, z(3), 1.0, torch.eye(4), torch.eye(4), 0, z(3), False, False)
    with pytest.raises(CsplatError):
        GaussianRasterizer(rs)(means3D=z(4, 3), means2D=z(4, 3), opacities=z(4, 1), shs=z(4, 1, 3), scales=z(4, 3),
                               rotations=z(4, 4))
    with pytest.raises(CsplatError):
        distCUDA2(z(8, 3))
    with pytest.raises(Exception):   # argument validation mirrors upstream ("provide exactly one of ...")
        GaussianRasterizer(rs)(means3D=z(4, 3), means2D=z(4, 3), opacities=z(4, 1), shs=z(4, 1, 3), colors_precomp=z(4, 3),
                               scales=z(4, 3), rotations=z(4, 4))


def test_rotation_helpers():
    from csplat import rotations as r
    gen = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(50, 3, 3, generator=gen, dtype=torch.float64))[0]
    Q = Q * torch.det(Q)[:, None, None]
    x = torch.randn(50, 3, 3, generator=gen, dtype=torch.float64)  # a triangle per item
    y = x @ Q.transpose(1, 2) + torch.randn(50, 1, 3, generator=gen, dtype=torch.float64)
    R, t = r.rigid_points_registration(x, y)
    assert float((R @ x.transpose(1, 2) + t[:, :, None] - y.transpose(1, 2)).abs().max()) < 1e-9
    q = r.rotmat_to_unitquat(Q)
    # xyzw quaternion -> matrix round trip
    X, Y, Z, Wq = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rq = torch.stack([1 - 2 * (Y * Y + Z * Z), 2 * (X * Y - Z * Wq), 2 * (X * Z + Y * Wq),
                      2 * (X * Y + Z * Wq), 1 - 2 * (X * X + Z * Z), 2 * (Y * Z - X * Wq),
                      2 * (X * Z - Y * Wq), 2 * (Y * Z + X * Wq), 1 - 2 * (X * X + Y * Y)], 1).reshape(-1, 3, 3)
    assert float((Rq - Q).abs().max()) < 1e-9
    ident = torch.tensor([[0.0, 0, 0, 1]], dtype=torch.float64).expand(50, 4)
    assert float((r.quat_composition([q, ident]) - q).abs().max()) < 1e-12


def test_separable_ssim_equals_2d_window_ssim():
    """csplat.train.ssim (two 11-tap passes) == the reference's 11x11 grouped-conv formulation (utils/loss_utils.py:30-70,
    restated inline: that module imports lpips and cannot be imported here)."""
    from math import exp
    import torch.nn.functional as F
    from csplat import train as tr
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(2, 3, 40, 52, generator=g), torch.rand(2, 3, 40, 52, generator=g)
    w1 = torch.tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    w1 = (w1 / w1.sum()).unsqueeze(1)
    w = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous()
    c = lambda x: F.conv2d(x, w, padding=5, groups=3)  # noqa: E731
    mu1, mu2 = c(a), c(b)
    s1, s2, s12 = c(a * a) - mu1 ** 2, c(b * b) - mu2 ** 2, c(a * b) - mu1 * mu2
    ref = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))
    assert abs(float(ref.mean()) - float(tr.ssim(a, b))) < 1e-6
    assert float((ref - tr.ssim(a, b, return_map=True)).abs().max()) < 1e-5
    assert abs(float(tr.ssim(a, a)) - 1.0) < 1e-5


def test_closed_form_triangle_kabsch_equals_svd():
    """csplat.rotations.kabsch_triangles (no SVD) == the SVD Kabsch solution, values and gradients, incl. non-rigid
    deformations and mirrored triangles (determinant fix)."""
    from csplat import rotations as r
    g = torch.Generator().manual_seed(0)
    x = torch.randn(500, 3, 3, generator=g, dtype=torch.float64)
    Q = torch.linalg.qr(torch.randn(500, 3, 3, generator=g, dtype=torch.float64))[0]
    Q = Q * torch.det(Q)[:, None, None]
    y = x @ Q.transpose(1, 2) + torch.randn(500, 1, 3, generator=g, dtype=torch.float64) \
        + 0.2 * torch.randn(500, 3, 3, generator=g, dtype=torch.float64)
    y[:50] = -y[:50]
    y.requires_grad_(True)
    R1, _ = r.rigid_points_registration(x, y)
    R2 = r.kabsch_triangles(x, y)
    assert float((R1 - R2).abs().max()) < 1e-10 and float((torch.det(R2) - 1).abs().max()) < 1e-10
    w = torch.randn(500, 3, 3, generator=g, dtype=torch.float64)
    g1, = torch.autograd.grad((R1 * w).sum(), y, retain_graph=True)
    g2, = torch.autograd.grad((R2 * w).sum(), y)
    assert float((g1 - g2).abs().max() / g1.abs().max()) < 1e-9


# ------------------------------------------------------------------ the C-ABI library loads and exports the header
def test_cabi_exports_every_declared_symbol():
    from csplat import native
    hdr = open(os.path.join(util.ROOT, "include", "csplat.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(csplat_[a-z0-9_]+)\s*\(", hdr)) - {"csplat_alloc_fn"}
    assert len(declared) >= 18
    lib = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libcsplat.so does not export {name}"
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    assert native.lib.csplat_abi_version() == native.ABI_VERSION
    # size / layout helpers are pure host code
    assert native.lib.csplat_geom_bytes(1000) > 1000 * (4 + 8 + 16 + 12 + 24)
    o3 = (C.c_size_t * 3)()
    native.lib.csplat_image_layout(800, 800, o3)
    assert o3[0] == 0 and o3[1] >= 2500 * 8 and o3[2] - o3[1] >= 640000 * 4


def test_densify_prune_and_adam_surgery_replay_the_reference():
    """csplat/densify.py against the reference's own MultiGaussianMesh run (tests/golden/densify.npz, generated by
    make_golden.gen_densify on CPU tensors): three Adam steps, two rounds of statistics, densify (clone + split, same
    global RNG seed), prune with a screen-size limit, opacity reset, one more Adam step -- parameters, Adam moments, step
    counters, face ids and statistics identical after every stage, for torch.optim.Adam and for GroupedAdam's state."""
    import types
    import torch
    from csplat.gaussians import MeshGaussians
    g = golden("densify.npz")
    T = lambda a: torch.tensor(a)  # noqa: E731
    pc = MeshGaussians(3)
    pc.mesh = types.SimpleNamespace(pos=T(g["pos"]), face=T(g["face"]), edge_index=None)
    pc.face_ids = T(g["face_ids"])
    names = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    attrs = ["face_bary", "face_offset", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    for n, a in zip(names, attrs):
        setattr(pc, a, torch.nn.Parameter(T(g["init." + n])))
    pc.fused = False
    lrs = [1.6e-4, 1.6e-4, 2.5e-3, 2.5e-3 / 20, 0.05, 0.005, 0.001]
    pc.optimizer = torch.optim.Adam([{"params": [getattr(pc, a)], "lr": lr, "name": n} for a, lr, n in zip(attrs, lrs, names)],
                                    lr=0.0, eps=1e-15)
    pc.densification_setup(percent_dense=0.01)
    pc.max_radii2D = T(g["max_radii2D"])

    def feed(flat):
        off = 0
        for grp in pc.optimizer.param_groups:
            p = grp["params"][0]
            p.grad = T(flat[off:off + p.numel()]).reshape(p.shape)
            off += p.numel()
        return off

    def check(tag):
        for grp in pc.optimizer.param_groups:
            p, n = grp["params"][0], grp["name"]
            st = pc.optimizer.state[p]
            assert p is getattr(pc, dict(zip(names, attrs))[n])
            np.testing.assert_array_equal(p.detach().numpy(), g[f"{tag}.{n}"])
            np.testing.assert_array_equal(st["exp_avg"].numpy(), g[f"{tag}.{n}.exp_avg"])
            np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), g[f"{tag}.{n}.exp_avg_sq"])
            assert float(st["step"]) == float(g[f"{tag}.{n}.step"])
        np.testing.assert_array_equal(pc.face_ids.numpy(), g[f"{tag}.face_ids"])
        np.testing.assert_array_equal(pc.pos_gradient_accum.numpy(), g[f"{tag}.pos_gradient_accum"])
        np.testing.assert_array_equal(pc.denom.numpy(), g[f"{tag}.denom"])
        np.testing.assert_array_equal(pc.max_radii2D.numpy(), g[f"{tag}.max_radii2D"])

    flat, per = g["adam_grads"], sum(getattr(pc, a).numel() for a in attrs)
    for it in range(3):
        feed(flat[it * per:(it + 1) * per])
        pc.optimizer.step()
    check("stepped")
    vsp, upd = T(g["vsp"]), T(g["update_filter"])
    pc.add_densification_stats(vsp, upd)
    pc.add_densification_stats(vsp * 0.5, upd)
    check("stats")
    torch.manual_seed(4321)
    pc.densify(2e-4, 0.05, 1.0, None)
    assert pc.face_bary.shape[0] == g["densified.face_ids"].shape[0] == 81
    check("densified")
    pc.prune(2e-4, 0.3, 1.0, 20)
    check("pruned")
    pc.reset_opacity()
    check("reset")
    feed(g["post_grads"])
    pc.optimizer.step()
    check("after_step")
    # the renderer-facing accessors still work on the re-created parameters
    assert pc.get_xyz().shape == (pc.face_ids.shape[0], 3) and pc.get_features.shape[1:] == (16, 3)


def test_ply_layout_and_round_trip(tmp_path):
    """csplat/ply.py: the byte layout plyfile produces for PlyElement.describe(float32 records, 'vertex') written with
    PlyData([el]).write(path) on a little-endian host (known-answer header + packed little-endian records), the
    reference's attribute order (gaussian_model.py:181-193 + b1,b2,b3,o,id), and MeshGaussians.save_ply -> load_ply."""
    import types
    import torch
    from csplat import ply
    from csplat.gaussians import MeshGaussians
    names = ["x", "y", "z"]
    ply.write_ply(tmp_path / "t.ply", names, np.array([[1.0, 2.0, 3.0], [-0.5, 0.25, 4.0]]))
    raw = (tmp_path / "t.ply").read_bytes()
    head = b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
    assert raw == head + np.array([1.0, 2.0, 3.0, -0.5, 0.25, 4.0], "<f4").tobytes()
    d = ply.read_ply(tmp_path / "t.ply")
    assert list(d) == names and d["y"].tolist() == [2.0, 0.25]
    (tmp_path / "a.ply").write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty uchar r\n"
                                    "end_header\n0.5 7\n1.5 9\n")
    d = ply.read_ply(tmp_path / "a.ply")
    assert d["x"].tolist() == [0.5, 1.5] and d["r"].dtype == np.uint8 and d["r"].tolist() == [7, 9]
    assert ply.attribute_names(3, 45) == (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] +
                                          [f"f_rest_{i}" for i in range(45)] + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0",
                                                                              "rot_1", "rot_2", "rot_3", "b1", "b2", "b3", "o", "id"])
    g = torch.Generator().manual_seed(1)
    V, F, P = 9, 8, 17
    pc = MeshGaussians(3)
    pc.fused = False
    pos = torch.rand(V, 3, generator=g)
    face = torch.stack([torch.randperm(V, generator=g)[:3] for _ in range(F)], 1)
    ei = torch.randint(0, V, (2, 20), generator=g)
    pc.mesh = types.SimpleNamespace(pos=pos, face=face, edge_index=ei)
    pc.face_ids = torch.randint(0, F, (P,), generator=g)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g))  # noqa: E731
    pc.face_bary, pc.face_offset = torch.nn.Parameter(torch.rand(P, 3, generator=g) + 0.1), mk(P, 1)
    pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation = mk(P, 1, 3), mk(P, 15, 3), mk(P, 1), mk(P, 3), mk(P, 4)
    pc.save_ply(str(tmp_path / "it"))
    d = ply.read_ply(tmp_path / "it" / "point_cloud.ply")
    assert len(d) == 6 + 3 + 45 + 1 + 3 + 4 + 5 and d["x"].shape == (P,)
    np.testing.assert_array_equal(np.stack([d["x"], d["y"], d["z"]], 1), pc.get_xyz().detach().numpy())
    np.testing.assert_array_equal(d["f_rest_1"], pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy()[:, 1])
    q = MeshGaussians(3)
    q.fused = False
    q.load_ply(str(tmp_path / "it"), device="cpu")
    for a in ("face_bary", "face_offset", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert isinstance(getattr(q, a), torch.nn.Parameter)
        np.testing.assert_array_equal(getattr(q, a).detach().numpy(), getattr(pc, a).detach().numpy())
    assert torch.equal(q.face_ids, pc.face_ids) and torch.equal(q.mesh.face, face) and torch.equal(q.mesh.pos, pos)
    np.testing.assert_array_equal(q.get_xyz().detach().numpy(), pc.get_xyz().detach().numpy())


def test_hdf5min_known_answer_structure_and_round_trip(tmp_path):
    """csplat/hdf5min.py (the `mesh.hdf5` side-car, gaussian_mesh.py:462-465 / data_utils.py:450-457): structural known answers
    from the HDF5 file-format specification -- signature, version-0 superblock fields, the root entry's cached B-tree / heap
    addresses landing on "TREE" / "HEAP", one "SNOD" with the datasets in name order, end-of-file address == file size, 8-byte
    aligned raw data -- and a bit-exact round trip of the four datasets the reference stores (float32 pos / norm, int64 face /
    edge_index), a float64 and an int32 array, a 0-d and an empty one.  (libhdf5 / h5py are not installed: interoperability with
    them is by construction from the specification, not tested.)"""
    import struct
    from csplat import hdf5min
    rng = np.random.default_rng(2)
    arrays = {"pos": rng.normal(size=(37, 3)).astype(np.float32), "norm": rng.normal(size=(37, 3)).astype(np.float32),
              "face": rng.integers(0, 37, (3, 50)).astype(np.int64), "edge_index": rng.integers(0, 37, (2, 120)).astype(np.int64),
              "f64": rng.normal(size=(4, 2, 3)), "i32": np.arange(-5, 6, dtype=np.int32), "scalar": np.array(3.5, np.float32),
              "empty": np.zeros((0, 3), np.float32)}
    path = str(tmp_path / "mesh.hdf5")
    hdf5min.save(path, arrays)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])
    leaf_k, internal_k, flags = struct.unpack_from("<HHI", raw, 16)
    assert (leaf_k, internal_k, flags) == (4, 16, 0)
    base, free, eof, drv = struct.unpack_from("<QQQQ", raw, 24)
    assert base == 0 and free == drv == 0xFFFFFFFFFFFFFFFF and eof == len(raw)
    name_off, hdr, ctype, _, btree, heap = struct.unpack_from("<QQIIQQ", raw, 56)
    assert ctype == 1 and raw[btree:btree + 4] == b"TREE" and raw[heap:heap + 4] == b"HEAP"
    assert raw[hdr] == 1 and struct.unpack_from("<H", raw, hdr + 2)[0] == 1                 # v1 object header, one message
    assert struct.unpack_from("<HH", raw, hdr + 16) == (0x0011, 16)                          # ... the symbol-table message
    snod = struct.unpack_from("<Q", raw, btree + 32)[0]
    assert raw[snod:snod + 4] == b"SNOD" and struct.unpack_from("<H", raw, snod + 6)[0] == len(arrays)
    heap_data = struct.unpack_from("<Q", raw, heap + 24)[0]
    names = []
    for e in range(len(arrays)):
        off = struct.unpack_from("<Q", raw, snod + 8 + 40 * e)[0]
        names.append(raw[heap_data + off:raw.index(b"\0", heap_data + off)].decode())
    assert names == sorted(arrays)
    # datatype messages: the byte strings libhdf5 emits for H5T_IEEE_F32LE / H5T_STD_I64LE (HDF5 File Format Specification, IV.A.2.d:
    # class + version, class bit fields, size; then bit offset / precision (+ exponent location 23, size 8, mantissa location 0, size
    # 23, bias 127 for the float))
    assert hdf5min._dtype_message(np.float32) == bytes.fromhex("11201f00040000000000200017080017" "7f000000")
    assert hdf5min._dtype_message(np.int64) == bytes.fromhex("1008000008000000" "00004000")
    assert hdf5min._dtype_message(np.float64) == bytes.fromhex("11203f000800000000004000340b0034" "ff030000")
    # the empty dataset has no storage: its layout message carries the undefined address (as libhdf5 writes it)
    e_hdr = struct.unpack_from("<Q", raw, snod + 8 + 40 * sorted(arrays).index("empty") + 8)[0]
    at, found = e_hdr + 16, None
    for _ in range(struct.unpack_from("<H", raw, e_hdr + 2)[0]):
        mtype, msize = struct.unpack_from("<HH", raw, at)
        if mtype == 0x0008:
            found = struct.unpack_from("<BBQQ", raw, at + 8)
        at += 8 + msize
    assert found == (3, 1, 0xFFFFFFFFFFFFFFFF, 0)
    back = hdf5min.load(path, prefer_h5py=False)
    assert set(back) == set(arrays)
    for k, v in arrays.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape
        np.testing.assert_array_equal(back[k], v)
    with pytest.raises(ValueError):
        hdf5min.save(path, {f"d{i}": np.zeros(1) for i in range(9)})
    open(path, "wb").write(b"not hdf5 at all")
    with pytest.raises(ValueError):
        hdf5min.load(path, prefer_h5py=False)


def test_cleanup_barycentric_coordinates_replays_the_reference():
    """the vectorised cleanup_barycentric_coordinates against the reference's per-Gaussian Python loop, run on CPU tensors
    by make_golden.gen_densify on a grid mesh with rows that have one, two and three negative coordinates: same faces,
    same coordinates (border rows included: the reference's 0.005 / 0.005 = 1.0 is reproduced)."""
    import types
    import torch
    from csplat.gaussians import MeshGaussians
    g = golden("densify.npz")
    T = lambda a: torch.tensor(a)  # noqa: E731
    pc = MeshGaussians(3)
    pc.fused = False
    pc.mesh = types.SimpleNamespace(pos=T(g["cleanup.pos"]), face=T(g["cleanup.face"]), edge_index=None)
    pc.face_ids = T(g["cleanup.face_ids_in"])
    pc.face_bary = torch.nn.Parameter(T(g["cleanup.face_bary_in"]))
    pc.cleanup_barycentric_coordinates()
    np.testing.assert_array_equal(pc.face_ids.numpy(), g["cleanup.face_ids_out"])
    np.testing.assert_allclose(pc.face_bary.detach().numpy(), g["cleanup.face_bary_out"], rtol=1e-6, atol=0)
    assert (g["cleanup.face_ids_out"] != g["cleanup.face_ids_in"]).sum() > 20 and (g["cleanup.face_bary_out"] == 1.0).any()
    before = pc.face_bary.detach().clone()
    clean = MeshGaussians(3); clean.fused = False
    clean.mesh, clean.face_ids = pc.mesh, pc.face_ids.clone()
    clean.face_bary = torch.nn.Parameter(before.abs())
    clean.cleanup_barycentric_coordinates()                       # nothing negative: untouched
    assert torch.equal(clean.face_bary.detach(), before.abs())


def test_blender_scene_loader_matches_reference():
    """csplat/scene_io.py against what the reference's readCamerasFromTransforms / read_timeline return for the committed
    tests/golden/blender_scene (tests/golden/scene_io.npz, generated by make_golden.py::gen_scene_io)."""
    import shutil
    pytest.importorskip("PIL")
    from csplat import scene_io as sio
    g = golden("scene_io.npz")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blender_scene")
    mapper, top = sio.read_timeline(root)
    np.testing.assert_array_equal(np.array(sorted(mapper)), g["timeline.keys"])
    np.testing.assert_array_equal(np.array([mapper[k] for k in sorted(mapper)]), g["timeline.values"])
    assert top == float(g["timeline.max"])

    def check(tag, infos):
        assert len(infos) == int(g[f"{tag}.n"])
        for i, c in enumerate(infos):
            np.testing.assert_allclose(c.R, g[f"{tag}.{i}.R"], rtol=0, atol=1e-15)
            np.testing.assert_allclose(c.T, g[f"{tag}.{i}.T"], rtol=0, atol=1e-15)
            np.testing.assert_array_equal(np.array([c.FovX, c.FovY]), g[f"{tag}.{i}.fov"])
            np.testing.assert_array_equal(c.image.numpy(), g[f"{tag}.{i}.image"])          # bit-exact composite + quantisation
            np.testing.assert_array_equal(np.array([c.uid, c.width, c.height, c.view_id, c.time_id]), g[f"{tag}.{i}.ints"])
            assert c.time == float(g[f"{tag}.{i}.time"]) and c.image_name == str(g[f"{tag}.{i}.name"])
            if f"{tag}.{i}.mask" in g.files:
                np.testing.assert_array_equal(c.mask.numpy(), g[f"{tag}.{i}.mask"])
            else:
                assert c.mask is None
    check("train_white", sio.read_cameras_from_transforms(root, "transforms_train.json", True))
    check("train_black_skip", sio.read_cameras_from_transforms(root, "transforms_train.json", False, time_skip=2, view_skip=2))
    # (the mask directory only covers the train names: the fixture was generated with it moved aside for the test split)
    tmp = os.path.join(os.path.dirname(root), "_scene_copy")
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.copytree(root, tmp, ignore=shutil.ignore_patterns("masks_gripper"))
    try:
        check("test_white", sio.read_cameras_from_transforms(tmp, "transforms_test.json", True))
        train, test, video, _, _ = sio.read_blender_scene(tmp, True)
        assert len(train) == 6 and len(test) == 4 and video is None
        cam = sio.camera_from_info(train[0], device="cpu")
        assert (cam.image_height, cam.image_width) == (6, 8) and cam.world_view_transform.shape == (4, 4)
        # the matrices follow the reference's Camera (pinned by camera.npz through csplat.synthetic.camera_matrices)
        assert abs(float(torch.det(cam.world_view_transform[:3, :3])) - 1.0) < 1e-5
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_grouped_adam_state_dict_leaves_the_live_state_alone():
    """torch's Optimizer.state_dict() hands out the per-parameter state dicts by reference; GroupedAdam.state_dict()
    must not replace the live moments (views into the capacity buffers of csplat/store.py) by compact clones -- the optimizer would go on
    updating the clones and the next compaction would scatter the store's stale rows back.  A checkpoint save between two steps of the
    densification phase must change nothing: state_dict() -> steps -> prune == the same run without state_dict(), bit for bit; and the
    saved copy is compact (no spare capacity in the checkpoint, gaussian_model.py:64-75)."""
    import types
    import torch
    from csplat.gaussians import MeshGaussians
    from csplat.optim import GroupedAdam
    g = golden("densify.npz")
    T = lambda a: torch.tensor(a)  # noqa: E731
    names = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    attrs = ["face_bary", "face_offset", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    lrs = [1.6e-4, 1.6e-4, 2.5e-3, 2.5e-3 / 20, 0.05, 0.005, 0.001]

    def run(save):
        pc = MeshGaussians(3)
        pc.mesh = types.SimpleNamespace(pos=T(g["pos"]), face=T(g["face"]), edge_index=None)
        pc.face_ids = T(g["face_ids"])
        for n, a in zip(names, attrs):
            setattr(pc, a, torch.nn.Parameter(T(g["init." + n])))
        pc.fused = False
        pc.optimizer = GroupedAdam([{"params": [getattr(pc, a)], "lr": lr, "name": n} for a, lr, n in zip(attrs, lrs, names)], lr=0.0, eps=1e-15)
        pc.densification_setup(percent_dense=0.01)
        pc.max_radii2D = T(g["max_radii2D"])
        gen = torch.Generator().manual_seed(3)

        def step():
            for grp in pc.optimizer.param_groups:
                p = grp["params"][0]
                p.grad = torch.randn(p.shape, generator=gen)
            pc.optimizer.step()
        for _ in range(3):
            step()
        vsp, upd = T(g["vsp"]), T(g["update_filter"])
        pc.add_densification_stats(vsp, upd)
        torch.manual_seed(4321)
        pc.densify(2e-4, 0.05, 1.0, None)            # from here on every moment is a view into a capacity buffer
        step()
        sd = None
        if save:
            live = {n: pc.optimizer.state[grp["params"][0]]["exp_avg"] for grp, n in zip(pc.optimizer.param_groups, names)}
            sd = pc.optimizer.state_dict()
            for grp, n in zip(pc.optimizer.param_groups, names):        # the live state is untouched ...
                assert pc.optimizer.state[grp["params"][0]]["exp_avg"] is live[n]
            for st in sd["state"].values():                             # ... and the saved one is compact
                for k in ("exp_avg", "exp_avg_sq"):
                    assert st[k].untyped_storage().nbytes() == st[k].numel() * st[k].element_size()
        for _ in range(5):
            step()
        pc.prune(2e-4, 0.3, 1.0, 20)
        step()
        out = {}
        for grp, n in zip(pc.optimizer.param_groups, names):
            p = grp["params"][0]
            st = pc.optimizer.state[p]
            out[n] = (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), float(st["step"]))
        return out, sd
    a, sd = run(True)
    b, _ = run(False)
    for n in names:
        for x, y in zip(a[n][:3], b[n][:3]):
            assert torch.equal(x, y), n
        assert a[n][3] == b[n][3]
    # the saved state loads into a fresh optimizer of the same shape (torch's own contract)
    fresh = GroupedAdam([{"params": [torch.nn.Parameter(torch.zeros_like(sd["state"][i]["exp_avg"]))], "lr": lr, "name": n}
                         for i, (lr, n) in enumerate(zip(lrs, names))], lr=0.0, eps=1e-15)
    fresh.load_state_dict(sd)


H5PY_PYTHON = "/opt/conda/bin/python3.9"          # an interpreter WITH the real h5py (3.3.0 on libhdf5 1.10.6) on this image


def _real_h5py():
    import subprocess
    if not os.path.exists(H5PY_PYTHON):
        return False
    try:
        return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        return False


def test_hdf5min_reads_a_file_written_by_the_real_h5py():
    """N4 interop, direction 1 (always runs): tests/golden/mesh_h5py.hdf5 was written by the REAL h5py / libhdf5 with the reference's
    call pattern (gaussian_mesh.py:462-465; generator tests/golden/make_h5py_fixture.py) -- csplat/hdf5min.py's own reader must
    return its four datasets bit for bit (what meshnet/data_utils.py:450-457 reads back with h5py)."""
    from csplat import hdf5min
    ref = golden("mesh_h5py.npz")
    got = hdf5min.load(os.path.join(util.GOLDEN, "mesh_h5py.hdf5"), prefer_h5py=False)
    assert sorted(got) == sorted(ref.files) == ["edge_index", "face", "norm", "pos"]
    for k in ref.files:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape
        np.testing.assert_array_equal(got[k], ref[k])


@pytest.mark.skipif(not _real_h5py(), reason="no interpreter with the real h5py on this machine")
def test_hdf5min_round_trips_through_the_real_h5py(tmp_path):
    """N4 interop, both directions LIVE against the real h5py 3.3.0 / libhdf5 1.10.6 found in the image's conda environment (a
    separate interpreter: the build's Python has no h5py): (1) a file written by hdf5min.save -- the reference's four datasets plus a
    float64, an int32, a 0-d and an empty array -- is opened by h5py, which must list the same names, dtypes, shapes, contiguous
    (unchunked) storage and values; (2) a file h5py writes with the reference's call pattern is read by hdf5min's own reader."""
    import subprocess
    from csplat import hdf5min
    rng = np.random.default_rng(5)
    arrays = {"pos": rng.normal(size=(37, 3)).astype(np.float32), "norm": rng.normal(size=(37, 3)).astype(np.float32),
              "face": rng.integers(0, 37, (3, 50)).astype(np.int64), "edge_index": rng.integers(0, 37, (2, 120)).astype(np.int64),
              "f64": rng.normal(size=(4, 2, 3)), "i32": np.arange(-5, 6, dtype=np.int32), "scalar": np.array(3.5, np.float32),
              "empty": np.zeros((0, 3), np.float32)}
    ours, theirs, dump = str(tmp_path / "ours.hdf5"), str(tmp_path / "theirs.hdf5"), str(tmp_path / "dump.npz")
    hdf5min.save(ours, arrays)
    np.savez(str(tmp_path / "in.npz"), **arrays)
    code = ("import sys, h5py, numpy as np\n"
            "ours, theirs, dump, inp = sys.argv[1:5]\n"
            "out = {}\n"
            "with h5py.File(ours, 'r') as f:\n"
            "    for k in f.keys():\n"
            "        d = f[k]\n"
            "        assert d.chunks is None and d.compression is None, k\n"
            "        out[k] = d[()]\n"
            "np.savez(dump, **out)\n"
            "src = np.load(inp)\n"
            "with h5py.File(theirs, 'w') as f:\n"
            "    for k in src.files:\n"
            "        f.create_dataset(k, data=src[k])\n")
    r = subprocess.run([H5PY_PYTHON, "-c", code, ours, theirs, dump, str(tmp_path / "in.npz")], capture_output=True, text=True, timeout=300,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    seen = np.load(dump)
    assert sorted(seen.files) == sorted(arrays)
    for k, v in arrays.items():
        assert seen[k].dtype == v.dtype and seen[k].shape == v.shape, k
        np.testing.assert_array_equal(seen[k], v)
    back = hdf5min.load(theirs, prefer_h5py=False)
    assert sorted(back) == sorted(arrays)
    for k, v in arrays.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape, k
        np.testing.assert_array_equal(back[k], v)


def test_a_failed_call_drops_the_ticketed_scratch_caches():
    """csplat.native.check: an entry point that fails may have left a "zero on entry, left at zero" ticket word anywhere; every cache of
    such scratch registers in native.TICKET_CACHES and is emptied by ANY error, so the next call zeroes a fresh buffer."""
    from csplat import native
    cache = {"k": object()}
    native.TICKET_CACHES.append(cache)
    try:
        native.check(0, "fine")
        assert cache
        with pytest.raises(native.CsplatError):
            native.check(3, "csplat_something")
        assert not cache
    finally:
        native.TICKET_CACHES[:] = [c for c in native.TICKET_CACHES if c is not cache]     # (by identity: an emptied dict equals every empty cache)
    import csplat.train as tr
    from meshnet import graph_ops
    # (the modules' own binding: another test may have imported a second copy of csplat.native)
    assert any(c is tr._L1_SCRATCH for c in tr._n.TICKET_CACHES) and any(c is tr._IMG_SCRATCH for c in tr._n.TICKET_CACHES)
    assert any(c is graph_ops._SIMH_SCRATCH for c in graph_ops._n.TICKET_CACHES)


_FULL = {}


def _full_size_case():
    if not _FULL:
        P, W, H = 100_000, 800, 800
        sc = syn.scene_1(P=P, W=W, H=H, n_cams=4)
        case = dict(g=syn.gaussians_at(sc), cam=sc["cameras"][1], W=W, H=H, P=P, bg=sc["bg"], sh_degree=3)
        _FULL.update(case=case, o=oracle_forward(case))
    return _FULL["case"], _FULL["o"]


def test_preprocess_full_size_against_a_numpy_fp64_restatement():
    """K1 at BASELINE configs[1] size against a restatement written here in numpy float64 from the published algorithm (SURVEY Appendix
    A.1) -- view transform, near cull at 0.2, homogeneous divide with the 1e-7 guard, Sigma = (S R)^T (S R) from the UN-normalised
    quaternion, the 1.3 tan(fov) frustum clamp, J W Sigma W^T J^T + 0.3 I, conic, radius = ceil(3 sqrt(lambda_max)) with the
    max(0.1, mid^2 - det) guard, ndc2pix -- sharing no line with oracle/raster_ref.c or csplat_raster.hip.  The fp32 oracle must agree
    with it: depth / centre / conic to fp32 rounding, the cull decision everywhere, the integer radius everywhere except where
    3 sqrt(lambda_max) lies within fp32 rounding of an integer (counted: < 1e-4 of the Gaussians, off by exactly one)."""
    case, o = _full_size_case()
    g, cam, W, H = case["g"], case["cam"], case["W"], case["H"]
    f8 = lambda a: np.asarray(a, np.float64)  # noqa: E731
    p, s, q = f8(g["means3D"]), f8(g["scales"]), f8(g["rotations"])
    V, Pm = f8(cam["world_view_transform"]).reshape(4, 4), f8(cam["full_proj_transform"]).reshape(4, 4)
    ph = np.concatenate([p, np.ones((len(p), 1))], 1)
    pv, hom = ph @ V, ph @ Pm                                  # row-vector convention (scene_reconstruction/cameras.py:63-67)
    seen = pv[:, 2] > 0.2
    w_ = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * w_[:, None]
    px, py = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]           # (r, x, y, z), NOT normalised in the kernel
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    A = R * s[:, None, :]                                      # R diag(s)
    Sig = A @ A.transpose(0, 2, 1)
    tanx, tany = float(cam["tanfovx"]), float(cam["tanfovy"])
    fx, fy = W / (2 * tanx), H / (2 * tany)
    tz = np.where(seen, pv[:, 2], 1.0)
    tx = np.clip(pv[:, 0] / tz, -1.3 * tanx, 1.3 * tanx) * tz
    ty = np.clip(pv[:, 1] / tz, -1.3 * tany, 1.3 * tany) * tz
    zero = np.zeros_like(tz)
    J = np.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    T = J @ V[:3, :3].T
    cov = T @ Sig @ T.transpose(0, 2, 1)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    mid = 0.5 * (a + c)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    rad = np.ceil(3.0 * np.sqrt(lam))
    on = o.radii > 0
    # a Gaussian the oracle kept is seen here (the oracle may drop more: zero-area rectangles)
    assert not np.any(on & ~seen)
    k = np.nonzero(on)[0]
    assert np.abs(o.depth[k] - pv[k, 2]).max() <= 2e-6 * np.abs(pv[k, 2]).max()
    assert np.abs(o.xy[k, 0] - px[k]).max() <= 2e-3 and np.abs(o.xy[k, 1] - py[k]).max() <= 2e-3          # pixels, fp32 at |x| ~ 800
    con = np.stack([c / det, -b / det, a / det], 1)
    assert np.abs(o.conic_opacity[k, :3] - con[k]).max() <= 2e-5 * np.abs(con[k]).max()
    dr = o.radii[k].astype(np.int64) - rad[k].astype(np.int64)
    assert np.abs(dr).max() <= 1 and (dr != 0).mean() < 1e-4, (int(np.abs(dr).max()), float((dr != 0).mean()))
    near_int = np.abs(3.0 * np.sqrt(lam[k][dr != 0]) - np.round(3.0 * np.sqrt(lam[k][dr != 0])))
    assert np.all(near_int < 1e-3)                             # every disagreement is a rounding tie at an integer radius


def test_binning_full_size_against_a_vectorised_numpy_restatement():
    """The independent binning restatement at BASELINE configs[1] size (the bit-exact index tests compare two
    compilations of one formula, and the independent restatement above runs on 600 Gaussians): P = 100,000, one 800 x 800 camera,
    R ~ 536k instances.  Rectangle from (centre, radius) by C-style truncation in float32, instances emitted tile-row-major per
    Gaussian in index order, key = tile << 32 | depth bits, numpy's STABLE argsort, ranges by run boundaries -- written with array
    operations, no line shared with oracle/raster_ref.c or the HIP kernels.  Keys, sorted ids, tiles touched and ranges identical.
    (The HIP path is held bit-exact to the same oracle output at this size by tests/test_raster_gpu.py::test_config2_full_size_*.)"""
    case, o = _full_size_case()
    P, W, H = case["P"], case["W"], case["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    vis = np.nonzero(o.radii > 0)[0]
    x, y, r = o.xy[vis, 0].astype(np.float32), o.xy[vis, 1].astype(np.float32), o.radii[vis].astype(np.float32)
    s16, s15 = np.float32(16), np.float32(15)
    trunc = lambda v: v.astype(np.int64)                       # C (int) cast of a float32: truncation towards zero  # noqa: E731
    minx = np.clip(trunc((x - r) / s16), 0, gx); maxx = np.clip(trunc((x + r + s15) / s16), 0, gx)
    miny = np.clip(trunc((y - r) / s16), 0, gy); maxy = np.clip(trunc((y + r + s15) / s16), 0, gy)
    nx, ny = maxx - minx, maxy - miny
    touched = nx * ny
    full = np.zeros(P, np.int64); full[vis] = touched
    np.testing.assert_array_equal(full, o.tiles_touched.astype(np.int64))
    keep = touched > 0
    vis, minx, miny, nx, touched = vis[keep], minx[keep], miny[keep], nx[keep], touched[keep]
    owner = np.repeat(np.arange(len(vis)), touched)             # instance -> its Gaussian (position in `vis`)
    local = np.arange(int(touched.sum())) - np.repeat(np.cumsum(touched) - touched, touched)
    ty = miny[owner] + local // nx[owner]                       # tile rows outer, columns inner
    tx = minx[owner] + local % nx[owner]
    dbits = o.depth[vis].astype(np.float32).view(np.uint32).astype(np.uint64)
    keys = ((ty * gx + tx).astype(np.uint64) << np.uint64(32)) | dbits[owner]
    ids = vis[owner].astype(np.uint32)
    assert len(keys) == o.R
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(keys[order], o.keys)
    np.testing.assert_array_equal(ids[order], o.ids)
    tiles = (keys[order] >> np.uint64(32)).astype(np.int64)
    first = np.r_[True, tiles[1:] != tiles[:-1]]
    starts = np.nonzero(first)[0]
    ends = np.r_[starts[1:], len(tiles)]
    ranges = np.zeros((gx * gy, 2), np.int32)
    ranges[tiles[starts], 0] = starts; ranges[tiles[starts], 1] = ends
    np.testing.assert_array_equal(ranges, o.ranges)


@pytest.mark.parametrize("E,N", [(0, 3), (1, 1), (5, 3), (64, 2), (1000, 37), (4099, 5000)])
def test_piece_numbering_of_the_fused_aggregation(E, N):
    """meshnet.graph_ops.piece_numbering (host side of csplat_gnn_edge_mlp3's fused aggregation, /root/reference/meshnet/graph_network.py:
    201-222 aggr = 'add'): against a plain loop -- a piece per maximal run of equal destination, cut every 8 rows, numbered in row order --
    and the property the node side relies on: summing rows by piece, then a node's pieces pp[v] .. pp[v + 1] - 1, gives the per-node sums
    of the rows (nodes without edges: empty ranges)."""
    import torch
    from meshnet.graph_ops import piece_numbering
    gen = torch.Generator().manual_seed(E + N)
    dst = torch.sort(torch.randint(0, N, (E,), generator=gen)).values
    if E > 40:
        dst[3:40] = dst[3]                     # a run across several groups of 8
        dst = torch.sort(dst).values
    rowptr = torch.zeros(N + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=N), 0).to(torch.int32)
    gp0, pp, npieces = piece_numbering(dst, rowptr)
    piece_of_row, p = [], -1
    for r in range(E):
        if r % 8 == 0 or dst[r] != dst[r - 1]:
            p += 1
        piece_of_row.append(p)
    assert npieces == p + 1 and gp0.tolist() == piece_of_row[0::8] and pp.shape == (N + 1,) and int(pp[-1]) == npieces
    vals = torch.randn(E, 3, generator=gen, dtype=torch.float64)
    pieces = torch.zeros(max(npieces, 1), 3, dtype=torch.float64).index_add_(0, torch.tensor(piece_of_row, dtype=torch.long), vals)
    per_node = torch.stack([pieces[int(pp[v]):int(pp[v + 1])].sum(0) for v in range(N)])
    ref = torch.zeros(N, 3, dtype=torch.float64).index_add_(0, dst, vals)
    assert torch.allclose(per_node, ref, atol=1e-12)


def test_rollout_real_world_refinement_host_form_replays_the_reference():
    """meshnet.rollout.refine_edge_lengths on CPU tensors (the host form: torch autograd + a fresh Adam, ten iterations) against the
    reference's own `real_world` branch (train_meshnet_sim.py:212-250, its text exec'd by tests/golden/make_golden.py: refine.npz),
    including the `length_deviation[grasped_particle] *= 0` entry; the grasped node is pinned afterwards as the rollout does."""
    import torch
    from meshnet.rollout import refine_edge_lengths
    d = util.golden("refine.npz")
    for name in ("a", "b"):
        t = lambda k: torch.from_numpy(d[f"{name}.{k}"])  # noqa: E731
        grasped = int(d[f"{name}.grasped"])
        v = refine_edge_lengths(t("pos"), t("v"), t("edge_index"), t("rest_len"), grasped)
        v[grasped] = t("action")
        ref = t("v_refined")
        assert float((v - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), name
