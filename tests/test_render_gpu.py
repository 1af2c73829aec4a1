"""gaussian_renderer.render() drop-in (boundary a1): signature, RenderResults record, gradients reaching the
simulator and the Gaussian parameters, and agreement of the whole render with the oracle given the same
rasterizer-level inputs."""
import math
from types import SimpleNamespace

import numpy as np
import pytest

import util
from util import image_err, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from csplat import synthetic as syn  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402


def _scene(P=3000, W=160, H=120, grid=14, n_times=5):
    sc = syn.scene_1(P=P, W=W, H=H, n_cams=1, grid=grid, n_times=n_times, seed=31)
    sc["log_scales"] = sc["log_scales"] + math.log(2.5)
    return sc


def _camera(c, time):
    t = lambda a: torch.tensor(a)  # noqa: E731
    return SimpleNamespace(image_height=c["image_height"], image_width=c["image_width"], FoVx=c["FoVx"], FoVy=c["FoVy"],
                           world_view_transform=t(c["world_view_transform"]), full_proj_transform=t(c["full_proj_transform"]),
                           camera_center=t(c["camera_center"]), time=time)


def _build(sc, dev="cuda"):
    from csplat.gaussians import MeshGaussians
    from meshnet.meshnet_network import ResidualMeshSimulator
    T = lambda a, dt=torch.float32: torch.tensor(a, device=dev, dtype=dt)  # noqa: E731
    pc = MeshGaussians(3).from_arrays(T(sc["mesh_pos"][0]), T(sc["faces"].T.copy(), torch.long), T(sc["edge_index"], torch.long),
                                      T(sc["face_ids"], torch.long), T(sc["bary"]), T(sc["log_scales"]), T(sc["quats"]),
                                      T(sc["opacity_logits"]), T(sc["sh"]))
    pc.active_sh_degree = 3
    sim = ResidualMeshSimulator(T(sc["mesh_pos"]), device=dev)
    return pc, sim


def test_render_results_and_gradients():
    from gaussian_renderer import render, RenderResults
    sc = _scene()
    pc, sim = _build(sc)
    with torch.no_grad():
        torch.manual_seed(0)
        sim.output.weight.normal_(0, 1e-3)
    cam = _camera(sc["cameras"][0], time=0.5)
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    bg = torch.ones(3, device="cuda")
    res = render(cam, pc, sim, pipe, bg, project_vertices=True)
    assert isinstance(res, RenderResults) and len(res) == 14
    P, V = sc["face_ids"].shape[0], sc["mesh_pos"].shape[1]
    assert res.render.shape == (3, 120, 160) and res.depth.shape == (1, 120, 160)
    assert res.radii.dtype == torch.int32 and res.radii.shape == (P,)
    assert res.visibility_filter.dtype == torch.bool and torch.equal(res.visibility_filter, res.radii > 0)
    assert res.viewspace_points.shape == (P, 3) and res.means3D_deform.shape == (P, 3) and res.rotations.shape == (P, 4)
    assert res.vertice_deform.shape == (V, 3) and res.projections.shape == (P, 2) and res.vertice_projections.shape == (V, 2)
    assert res.shadows is None and res.shadows_mean is None and res.shadows_std is None
    loss = (res.render - 0.5).abs().mean()
    loss.backward()
    assert res.viewspace_points.grad is not None and float(res.viewspace_points.grad[:, :2].abs().max()) > 0
    for p in (pc.face_bary, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0
    assert sim.output.weight.grad is not None and float(sim.output.weight.grad.abs().max()) > 0   # reaches the simulator
    # projections agree with the rasterizer's own pixel centres for visible Gaussians
    o = ro.forward(res.means3D_deform.detach().cpu().numpy(), pc.get_opacity.detach().cpu().numpy(),
                   sc["cameras"][0]["world_view_transform"], sc["cameras"][0]["full_proj_transform"],
                   sc["cameras"][0]["camera_center"], sc["cameras"][0]["tanfovx"], sc["cameras"][0]["tanfovy"], 160, 120,
                   np.ones(3), shs=pc.get_features.detach().cpu().numpy(), sh_degree=3,
                   scales=pc.get_scaling.detach().cpu().numpy(), rotations=res.rotations.detach().cpu().numpy(),
                   dtype=np.float64)
    assert image_err(res.render.detach().cpu().numpy(), o.color) < 1e-4        # whole render == oracle on same inputs
    assert image_err(res.depth.detach().cpu().numpy(), o.out_depth) < 1e-4
    np.testing.assert_array_equal(res.radii.cpu().numpy(), ro.forward(
        res.means3D_deform.detach().cpu().numpy(), pc.get_opacity.detach().cpu().numpy(),
        sc["cameras"][0]["world_view_transform"], sc["cameras"][0]["full_proj_transform"], sc["cameras"][0]["camera_center"],
        sc["cameras"][0]["tanfovx"], sc["cameras"][0]["tanfovy"], 160, 120, np.ones(3), shs=pc.get_features.detach().cpu().numpy(),
        sh_degree=3, scales=pc.get_scaling.detach().cpu().numpy(), rotations=res.rotations.detach().cpu().numpy(),
        stages="preprocess").radii)
    vis = o.radii > 0
    np.testing.assert_allclose(res.projections.detach().cpu().numpy()[vis], o.xy[vis], atol=2e-3)


def test_render_log_deform_path_writes_the_references_record(tmp_path):
    """render(log_deform_path=...) (gaussian_renderer/__init__.py:118-127): the debug dump with the reference's five arrays, the
    vertex rotations through MeshGaussians.get_vertice_rotation (gaussian_mesh.py:190-201; VERDICT r3 missing 6: the build's own
    Gaussian class used to lack it) -- and the render itself unchanged by the logging."""
    from gaussian_renderer import render
    sc = _scene(P=1000, W=64, H=64, grid=8)
    pc, sim = _build(sc)
    with torch.no_grad():
        torch.manual_seed(0)
        sim.output.weight.normal_(0, 1e-2)
    cam = _camera(sc["cameras"][0], time=0.5)
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    bg = torch.ones(3, device="cuda")
    path = str(tmp_path / "deform.npz")
    with torch.no_grad():
        plain = render(cam, pc, sim, pipe, bg)
        logged = render(cam, pc, sim, pipe, bg, log_deform_path=path)
    assert torch.equal(plain.render, logged.render)
    d = np.load(path)
    P, V = sc["face_ids"].shape[0], sc["mesh_pos"].shape[1]
    assert sorted(d.files) == ["means3D", "means3D_deform", "rotations", "vertice_deform", "vertice_rotations"]
    assert d["means3D"].shape == (P, 3) and d["means3D_deform"].shape == (P, 3) and d["rotations"].shape == (P, 4)
    assert d["vertice_deform"].shape == (V, 3) and d["vertice_rotations"].shape == (V, 4)
    np.testing.assert_array_equal(d["means3D_deform"], logged.means3D_deform.cpu().numpy())
    np.testing.assert_array_equal(d["means3D"], pc.get_xyz().detach().cpu().numpy())
    # the same quaternions from CPU float64 tensors of the same mesh
    pc_c, _ = _build(sc, dev="cpu")
    q64 = pc_c.get_vertice_rotation(torch.tensor(d["vertice_deform"], dtype=torch.float32)).numpy()
    ok = np.abs(q64[:, 3]) < 0.99999          # (near-identity rotations: the axis is a 0/0 in fp32 on both sides)
    np.testing.assert_allclose(d["vertice_rotations"][ok], q64[ok], atol=1e-4)
    np.testing.assert_allclose(np.linalg.norm(d["vertice_rotations"][ok], axis=1), 1.0, atol=1e-5)


def test_render_static_no_grad_and_override_color():
    from gaussian_renderer import render
    sc = _scene(P=1000, W=64, H=64, grid=8)
    pc, sim = _build(sc)
    cam = _camera(sc["cameras"][0], time=0.0)
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        a = render(cam, pc, sim, pipe, bg, render_static=True)
        col = torch.rand(1000, 3, device="cuda")
        b = render(cam, pc, sim, pipe, bg, render_static=True, override_color=col)
        pipe2 = SimpleNamespace(compute_cov3D_python=True, convert_SHs_python=False, debug=False)
        # the reference's python-covariance branch hands the rasterizer rotations AND cov3D_precomp (gaussian_renderer/__init__.py:
        # 83-88,156-164; recorded in tests/golden/render_wiring.npz), which upstream's wrapper rejects: same error here
        with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
            render(cam, pc, sim, pipe2, bg, render_static=True)
        # the covariance itself (gaussian_model.py:27-33) through the rasterizer's own cov3D_precomp input == the scale/rotation path
        from diff_gaussian_rasterization import GaussianRasterizer
        import gaussian_renderer as gr
        st, kw, _ = gr._prepare(cam, pc, sim, pipe, bg, 1.0, None, None, True)
        kw.update(scales=None, rotations=None, cov3D_precomp=pc.get_covariance(1.0))
        c_img = GaussianRasterizer(st)(**kw)[0]
    assert torch.isfinite(a.render).all() and a.vertice_projections is None
    assert not torch.allclose(a.render, b.render)
    assert image_err(c_img.cpu().numpy(), a.render.cpu().numpy()) < 1e-4       # python covariance == scale/rot path
    with pytest.raises(ValueError):
        render(_camera(sc["cameras"][0], time=1.4), pc, sim, pipe, bg)          # time beyond the mesh table


def test_from_mesh_uses_dist2():
    from csplat.gaussians import MeshGaussians
    sc = _scene(P=10, grid=10)
    T = lambda a, dt=torch.float32: torch.tensor(a, device="cuda", dtype=dt)  # noqa: E731
    pc = MeshGaussians(3).from_mesh(T(sc["mesh_pos"][0]), T(sc["faces"].T.copy(), torch.long), T(sc["edge_index"], torch.long),
                                    gaussian_init_factor=2, generator=torch.Generator(device="cuda").manual_seed(0))
    P = 2 * sc["faces"].shape[0]
    assert pc.num_gaussians == P and pc._scaling.shape == (P, 3)
    xyz = pc.get_xyz().detach().cpu().numpy()
    ref = np.log(np.sqrt(np.maximum(ro.dist2(xyz), 1e-7)))
    np.testing.assert_allclose(pc._scaling.detach().cpu().numpy()[:, 0], ref, rtol=1e-5, atol=1e-6)


def test_fused_mesh_transform_equals_torch_formulation():
    """csplat_mesh_transform_fwd/_bwd (closed-form Kabsch, hand-written adjoint) == the torch formulation of
    get_xyz / get_rotation (gaussian_mesh.py:151-188 with roma restated in csplat/rotations.py), values and gradients
    w.r.t. the deformed vertices, the barycentric weights and the rotation parameter."""
    sc = _scene(P=5000, W=64, H=64, grid=12)
    outs = []
    for fused in (True, False):
        pc, sim = _build(sc)
        pc.fused = fused
        torch.manual_seed(3)
        with torch.no_grad():
            pc._rotation.copy_(torch.randn_like(pc._rotation))           # non-trivial, un-normalised own rotations
        V = sc["mesh_pos"].shape[1]
        verts = (pc.mesh.pos + 0.05 * torch.randn(V, 3, device="cuda")).requires_grad_(True)
        xyz, rot = pc.get_xyz(verts), pc.get_rotation(verts)
        w1, w2 = torch.randn_like(xyz), torch.randn_like(rot)
        ((xyz * w1).sum() + (rot * w2).sum()).backward()
        outs.append((xyz.detach(), rot.detach(), verts.grad.clone(), pc.face_bary.grad.clone(), pc._rotation.grad.clone()))
    names = ("xyz", "rotation", "d_vertices", "d_bary", "d_rotation")
    for n, a, b in zip(names, outs[0], outs[1]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert err < 2e-4, (n, err)       # fp32 both sides; vertex grads: atomics vs sort-based index_put
    assert float((outs[0][1].norm(dim=1) - 1).abs().max()) < 1e-5


def test_mesh_transform_of_all_cameras_equals_per_camera():
    """transform_views([T,V,3]) (one launch each way) == get_xyz / get_rotation per camera: values, and the gradients on the
    deformed vertices of every camera and on the (shared) barycentric weights / rotation parameters summed over cameras."""
    sc = _scene(P=5000, W=64, H=64, grid=12)
    V = sc["mesh_pos"].shape[1]
    res = []
    for batched in (True, False):
        pc, sim = _build(sc)
        torch.manual_seed(5)
        with torch.no_grad():
            pc._rotation.copy_(torch.randn_like(pc._rotation))
        verts = (pc.mesh.pos[None] + 0.05 * torch.randn(3, V, 3, device="cuda")).requires_grad_(True)
        if batched:
            xyz, rot = pc.transform_views(verts)
        else:
            xyz, rot = zip(*[(pc.get_xyz(verts[t]), pc.get_rotation(verts[t])) for t in range(3)])
        ws = [torch.randn(5000, 7, device="cuda", generator=torch.Generator(device="cuda").manual_seed(t)) for t in range(3)]
        sum((xyz[t] * ws[t][:, :3]).sum() + (rot[t] * ws[t][:, 3:]).sum() for t in range(3)).backward()
        res.append((torch.stack(xyz).detach(), torch.stack(rot).detach(), verts.grad.clone(), pc.face_bary.grad.clone(),
                    pc._rotation.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])          # same kernel, same bits
    for n, a, b in zip(("d_vertices", "d_bary", "d_rotation"), res[0][2:], res[1][2:]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert err < 1e-5, (n, err)        # (atomic order / sum over cameras in registers vs autograd accumulation)


def test_render_views_equals_render_per_camera():
    """gaussian_renderer.render_views == [render(c) for c]: identical images / radii / depth, and the same gradients on
    the Gaussian parameters and the simulator (shared activations get one accumulated gradient buffer)."""
    from gaussian_renderer import render, render_views
    sc = _scene()
    pc, sim = _build(sc)
    with torch.no_grad():
        torch.manual_seed(0)
        sim.output.weight.normal_(0, 1e-3)
    cams = [_camera(sc["cameras"][0], time=t) for t in (0.25, 0.5, 0.75)]
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    bg = torch.ones(3, device="cuda")
    plist = [pc.face_bary, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation, sim.output.weight]

    def run(batched):
        for p in plist:
            p.grad = None
        res = render_views(cams, pc, sim, pipe, bg) if batched else [render(c, pc, sim, pipe, bg) for c in cams]
        sum((r.render - 0.4).abs().mean() for r in res).backward()
        return res, [p.grad.clone() for p in plist], [r.viewspace_points.grad.clone() for r in res]

    r1, g1, v1 = run(False)
    r2, g2, v2 = run(True)
    for a, b in zip(r1, r2):
        assert torch.equal(a.render, b.render) and torch.equal(a.radii, b.radii) and torch.equal(a.depth, b.depth)
        assert torch.equal(a.visibility_filter, b.visibility_filter)
    for a, b in zip(g1 + v1, g2 + v2):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    assert render_views([], pc, sim, pipe, bg) == []


def test_projection_kernel_matches_reference_formula_and_has_a_gradient():
    """csplat_project_points vs the reference's projections() (gaussian_renderer/__init__.py:166-179) restated in fp64."""
    import gaussian_renderer as gr
    from types import SimpleNamespace
    from csplat import synthetic as syn
    dev = torch.device("cuda")
    for W, H in ((800, 800), (640, 360)):
        c = syn.make_camera(33.0, W, H)
        full = torch.tensor(c["full_proj_transform"], device=dev)
        cam = SimpleNamespace(full_proj_transform=full, image_width=W, image_height=H)
        pts = (torch.rand(5001, 3, device=dev) - 0.5).requires_grad_()
        got = gr._project(cam, pts)
        assert type(got.grad_fn).__name__.startswith("_ProjectPoints")
        p64 = pts.detach().double().requires_grad_()
        hom = (full.double().T @ torch.cat([p64, torch.ones_like(p64[:, :1])], 1).T)
        ndc = (hom / hom[3, :])[:2].T
        ref = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
        assert float((got.double() - ref).abs().max()) < 2e-4 * max(W, H) / 800            # pixels
        w = torch.randn_like(got)
        got.backward(w)
        ref.backward(w.double())
        assert float((pts.grad.double() - p64.grad).abs().max()) < 1e-4 * float(p64.grad.abs().max())


WIRING_CASES = {"default": {}, "scale_mod": dict(scaling_modifier=1.7), "override_color": dict(override_color=True),
                "static": dict(render_static=True), "project_vertices": dict(project_vertices=True), "cov_python": dict(pipe_cov=True)}


@pytest.mark.parametrize("case", sorted(WIRING_CASES))
def test_render_wiring_equals_the_references_own_render(case):
    """What the build's render() puts into GaussianRasterizationSettings and hands the rasterizer, its call of the simulator and its
    by-products, against a RUN OF THE REFERENCE'S OWN render() (gaussian_renderer/__init__.py:39-206) under a recording stand-in for the
    rasterizer extension (tests/golden/make_golden.py:gen_render_wiring -> render_wiring.npz): tanfov = tan(FoV/2), bg,
    scale_modifier, the transposed matrices, sh_degree = the ACTIVE degree, campos; shs vs colors_precomp; scales vs cov3D_precomp;
    the deformed means and the composed rotations (roma served by scipy, fp64: quaternions equal up to sign); the zero screen-space
    tensor; projections / vertice_projections; which record fields are None."""
    import gaussian_renderer as gr
    from csplat.gaussians import MeshGaussians
    G = util.golden("render_wiring.npz")
    dev = torch.device("cuda")
    I = lambda k, dt=torch.float32: torch.tensor(G["in." + k], device=dev, dtype=dt)  # noqa: E731
    pos, face = I("pos"), I("face", torch.long)
    ei = torch.cat([face[[0, 1]], face[[1, 2]], face[[2, 0]]], 1)
    pc = MeshGaussians(3).from_arrays(pos, face, ei, I("face_ids", torch.long), I("face_bary"), I("scaling"), I("rotation"), I("opacity"),
                                      torch.cat([I("features_dc"), I("features_rest")], 1))
    pc.active_sh_degree = 2
    # the camera as the reference's Camera object presents itself to render(): these eight attributes
    cam = SimpleNamespace(image_height=int(G["in.cam_H"]), image_width=int(G["in.cam_W"]), FoVx=float(G["in.cam_FoVx"]),
                          FoVy=float(G["in.cam_FoVy"]), time=float(G["in.cam_time"]),
                          world_view_transform=torch.tensor(G["default.settings.viewmatrix"]),
                          full_proj_transform=torch.tensor(G["default.settings.projmatrix"]),
                          camera_center=torch.tensor(G["default.settings.campos"]))
    wave, seen, calls = I("wave"), [], []

    def simulator(time_vector):
        seen.append(time_vector.detach().clone())
        return pos + wave * time_vector

    class Recorder(gr.GaussianRasterizer):
        def forward(self, **kw):
            calls.append((self.raster_settings, kw))
            return super().forward(**kw)
    kw = dict(WIRING_CASES[case])
    pipe = SimpleNamespace(compute_cov3D_python=bool(kw.pop("pipe_cov", False)), convert_SHs_python=False, debug=False)
    if kw.get("override_color"):
        kw["override_color"] = I("override_color")
    real = gr.GaussianRasterizer
    gr.GaussianRasterizer = Recorder
    try:
        if case == "cov_python":
            with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
                gr.render(cam, pc, simulator, pipe, I("bg"), **kw)
            res = None
        else:
            res = gr.render(cam, pc, simulator, pipe, I("bg"), **kw)
    finally:
        gr.GaussianRasterizer = real
    assert len(calls) == 1
    rs, args = calls[0]
    g = lambda k: G[f"{case}.{k}"]  # noqa: E731
    close = lambda a, b, tol=1e-6: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) <= \
        tol * (float(np.abs(np.asarray(b, np.float64)).max()) + 1e-30)  # noqa: E731
    for f in ("image_height", "image_width", "sh_degree"):
        assert int(getattr(rs, f)) == int(g(f"settings.{f}")), f
    for f in ("prefiltered", "debug"):
        assert bool(getattr(rs, f)) == bool(g(f"settings.{f}")), f
    for f in ("tanfovx", "tanfovy", "scale_modifier"):
        assert abs(float(getattr(rs, f)) - float(g(f"settings.{f}"))) <= 1e-12 * abs(float(g(f"settings.{f}"))), f
    for f in ("bg", "viewmatrix", "projmatrix", "campos"):
        np.testing.assert_array_equal(getattr(rs, f).cpu().numpy(), g(f"settings.{f}"), err_msg=f)
        assert getattr(rs, f).is_cuda, f                      # (the reference calls .cuda() on the camera's matrices, :68-71)

    def same_quats(a, b, tol):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        sgn = np.sign((a * b).sum(1, keepdims=True))
        return float(np.abs(a * sgn - b).max()) <= tol
    for k in ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        assert (args.get(k) is None) == bool(g(f"arg.{k}.none")), k
        if args.get(k) is None:
            continue
        got = args[k].detach().cpu().numpy()
        assert got.shape == g(f"arg.{k}").shape, k
        if k == "rotations":
            assert same_quats(got, g("arg.rotations"), 5e-6)
        else:
            assert close(got, g(f"arg.{k}"), 2e-6), k
    assert float(args["means2D"].abs().max()) == 0.0 and args["means2D"].requires_grad
    assert len(seen) == int(g("sim_calls"))
    if seen:
        np.testing.assert_array_equal(seen[0].cpu().numpy(), g("sim_time_vector"))
    if res is None:
        return
    for f in res._fields:
        assert (getattr(res, f) is None) == bool(g(f"res.{f}.none")), f
    assert res.viewspace_points is args["means2D"] and res.viewspace_points.shape == g("res.viewspace_points").shape
    for f in ("means3D_deform", "vertice_deform", "opacities"):
        assert close(getattr(res, f).detach().cpu().numpy(), g(f"res.{f}"), 2e-6), f
    assert same_quats(res.rotations.detach().cpu().numpy(), g("res.rotations"), 5e-6)
    for f in ("projections", "vertice_projections"):
        if getattr(res, f) is not None:
            assert float(np.abs(getattr(res, f).detach().cpu().numpy() - g(f"res.{f}")).max()) < 2e-3, f       # pixels
    for f in ("render", "radii", "depth", "visibility_filter"):
        assert getattr(res, f).shape == g(f"res.{f}").shape and str(getattr(res, f).dtype).endswith(str(g(f"res.{f}").dtype)), f
