"""train_step analogue on a small scene: the whole hot path under autograd + two Adam
optimisers must fit a perturbed target: PSNR rises, loss falls, statistics have the reference's shapes."""
import contextlib

import numpy as np
import pytest

import util  # noqa: F401
from util import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_train_step_converges_small_scene():
    import bench_train as bt
    from csplat import train as tr
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    sc, pc, sim = bt.build(P=4000, W=160, H=128, grid=16, n_times=6, dev=dev)
    with torch.no_grad():
        pc._scaling.add_(0.9)      # ~2.5x larger splats so the 160x128 image is covered
    bg = torch.ones(3, device=dev)
    times = [0.2, 0.4, 0.6]
    with torch.no_grad():
        keep = pc._features_dc.detach().clone()
        torch.manual_seed(0)
        pc._features_dc.add_(0.5 * torch.randn_like(pc._features_dc))
        targets = [render(c, pc, sim, tr.DEFAULT_PIPE, bg).render.clamp(0, 1).clone() for c in bt.cameras(sc, times, dev)]
        pc._features_dc.copy_(keep)
    cams = bt.cameras(sc, times, dev, targets)
    pc.training_setup(feature_lr=0.01)
    mopt = torch.optim.Adam(sim.parameters(), lr=3e-4)
    ps, losses = [], []
    for it in range(1, 41):
        p, l, stats = tr.train_step(it, cams, pc, sim, mopt, background=bg)
        ps.append(float(p)); losses.append(float(l))
    P = pc.num_gaussians
    assert stats["viewspace_grad"].shape == (P, 3) and stats["radii"].shape == (P,) and stats["visibility_filter"].dtype == torch.bool
    assert all(torch.isfinite(q).all() for q in pc.parameters())
    assert ps[-1] > ps[0] + 1.0, (ps[0], ps[-1])           # >= 1 dB better after 40 steps
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])


def test_ssim_identity_and_psnr():
    from csplat import train as tr
    a = torch.rand(2, 3, 48, 40, device="cuda")
    assert abs(float(tr.ssim(a, a)) - 1.0) < 1e-5
    assert float(tr.ssim(a, torch.rand_like(a))) < 0.2
    b = (a + 0.1).clamp(0, 1)
    assert 15 < float(tr.psnr(a, b).mean()) < 25
    # csplat_psnr against the reference formula (utils/image_utils.py:17-21) in fp64, odd sizes and a misaligned batch stride
    for shape in ((2, 3, 48, 40), (3, 3, 37, 21), (1, 1, 5, 1), (4, 3, 200, 200)):
        x, y = torch.rand(*shape, device="cuda"), torch.rand(*shape, device="cuda")
        got = tr.psnr(x, y)
        mse = ((x.double() - y.double()) ** 2).view(shape[0], -1).mean(1, keepdim=True)
        ref = 20 * torch.log10(1.0 / torch.sqrt(mse))
        assert got.shape == ref.shape and float((got.double() - ref).abs().max()) < 1e-4, shape
        assert torch.equal(got, tr.psnr(x, y))          # fixed summation order


def test_blur_kernel_matches_grouped_conv_and_is_self_adjoint():
    """csplat_blur11 == the reference's zero-padded 11x11 grouped conv2d window (utils/loss_utils.py:30-58), on ragged
    sizes; gradient through it == blur of the gradient."""
    import torch.nn.functional as F
    from math import exp
    from csplat import train as tr
    g = torch.Generator(device="cuda").manual_seed(0)
    for (n, c, H, W) in [(2, 3, 37, 53), (1, 3, 128, 200), (3, 1, 16, 64)]:
        x = torch.rand(n, c, H, W, device="cuda", generator=g, requires_grad=True)
        w1 = torch.tensor([exp(-(k - 5) ** 2 / float(2 * 1.5 ** 2)) for k in range(11)])
        w1 = (w1 / w1.sum()).unsqueeze(1)
        w = w1.mm(w1.t()).double().unsqueeze(0).unsqueeze(0).expand(c, 1, 11, 11).contiguous()
        # the reference formulation runs on the CPU in fp64 (the GPU conv would go through MIOpen, whose grouped-conv
        # backward aborts intermittently on this image: nothing of ours, and not something a parity test should depend on)
        xc = x.detach().cpu().double().requires_grad_()
        ref = F.conv2d(xc, w, padding=5, groups=c)
        got = tr.GaussianBlur11.apply(x)
        assert float((ref.detach().float().cuda() - got).abs().max()) < 2e-6
        wgt = torch.rand_like(got)
        g_ref, = torch.autograd.grad((ref * wgt.cpu().double()).sum(), xc)
        g_got, = torch.autograd.grad((got * wgt).sum(), x)
        assert float((g_ref.float().cuda() - g_got).abs().max()) < 2e-6
    a, b = torch.rand(2, 3, 96, 80, device="cuda", generator=g), torch.rand(2, 3, 96, 80, device="cuda", generator=g)
    assert abs(float(tr.ssim(a, b)) - float(tr.ssim(a.cpu(), b.cpu()))) < 1e-6     # HIP window == torch conv formulation


@pytest.mark.parametrize("shape", [(3, 37, 53), (1,), (3, 800, 800), (7,)])
def test_fused_l1_matches_torch(shape):
    """csplat_l1 == torch.abs(a - b).mean() (utils/loss_utils.py:20-23) and its autograd gradient, ragged sizes, zeros
    (sign(0) = 0) and an upstream scale included; value to 1e-6 relative (different summation tree), gradient exact."""
    from csplat import train as tr
    g = torch.Generator(device="cuda").manual_seed(len(shape) + shape[0])
    a = torch.rand(*shape, device="cuda", generator=g, requires_grad=True)
    b = torch.rand(*shape, device="cuda", generator=g)
    with torch.no_grad():
        b.view(-1)[0] = a.view(-1)[0]          # an exact tie
    l1 = tr.l1_loss(a, b)
    (2.5 * l1).backward()
    g_fused, a.grad = a.grad.clone(), None
    ref = torch.abs(a - b).mean()
    (2.5 * ref).backward()
    assert abs(float(l1) - float(ref)) <= 1e-6 * abs(float(ref)) + 1e-12
    assert torch.equal(g_fused, a.grad)
    l2 = tr.l1_loss(a.detach(), b)              # no-grad path, and the scratch word was restored
    assert float(l2) == float(l1)


@pytest.mark.parametrize("shape", [(2, 3, 37, 53), (1, 3, 128, 200), (3, 1, 16, 64), (3, 3, 240, 200)])
def test_fused_ssim_matches_composed_formula(shape):
    """csplat_ssim_fwd/_bwd == the reference's SSIM formula (utils/loss_utils.py:40-70) evaluated in fp64 on the CPU with
    the grouped-conv window: value to 1e-6, gradient w.r.t. the rendered image to 1e-5 relative (ragged tile edges, zero
    padding of the images and of their products)."""
    import torch.nn.functional as F
    from math import exp
    from csplat import train as tr
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    a = torch.rand(*shape, device="cuda", generator=g, requires_grad=True)
    b = (a.detach() + 0.15 * torch.randn(*shape, device="cuda", generator=g)).clamp(0, 1)
    s = tr.ssim(a, b)
    assert type(s.grad_fn).__name__.startswith("FusedSSIM")
    (3.0 * s).backward()
    c = shape[1]
    w1 = torch.tensor([exp(-(k - 5) ** 2 / float(2 * 1.5 ** 2)) for k in range(11)])
    w1 = (w1 / w1.sum()).unsqueeze(1)                                        # float32 window, as the reference builds it
    w = w1.mm(w1.t()).double().unsqueeze(0).unsqueeze(0).expand(c, 1, 11, 11).contiguous()
    x, y = a.detach().cpu().double().requires_grad_(), b.cpu().double()
    cv = lambda t: F.conv2d(t, w, padding=5, groups=c)  # noqa: E731
    mu1, mu2 = cv(x), cv(y)
    s1, s2, s12 = cv(x * x) - mu1 * mu1, cv(y * y) - mu2 * mu2, cv(x * y) - mu1 * mu2
    ref = (((2 * mu1 * mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1 * mu1 + mu2 * mu2 + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))).mean()
    (3.0 * ref).backward()
    assert abs(float(s) - float(ref)) < 1e-6
    assert rel_err(a.grad.cpu().numpy(), x.grad.numpy()) < 1e-5
    with torch.no_grad():
        assert abs(float(tr.ssim(a, b)) - float(ref)) < 1e-6                 # no-grad path (no partials written)


def test_grouped_adam_matches_torch_adam():
    """csplat.optim.GroupedAdam (one HIP launch for all parameter groups) follows torch.optim.Adam: same parameters after
    6 steps (1e-6 relative), same state layout, per-group learning rates honoured, lr edits between steps, a parameter
    without gradient skipped; weight decay falls back to torch's implementation."""
    from csplat.optim import GroupedAdam
    gen = torch.Generator().manual_seed(4)
    shapes = [(1000, 3), (1000, 15, 3), (1000, 1), (7,), (1000, 4)]
    lrs = [1.6e-4, 2.5e-3 / 20, 0.05, 1e-3, 1e-3]
    init = [torch.randn(*s, generator=gen) for s in shapes]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        return ps, cls([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15, **kw)

    pa, oa = make(torch.optim.Adam)
    pb, ob = make(GroupedAdam)
    for it in range(6):
        for ps in (pa, pb):
            g2 = torch.Generator().manual_seed(100 + it)
            for k, p in enumerate(ps):
                p.grad = None if (k == 3 and it % 2 == 0) else (torch.randn(*p.shape, generator=g2) * 10.0 ** (k - 2)).cuda()
        if it == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-4
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert rel_err(y.detach().cpu().numpy(), x.detach().cpu().numpy()) < 1e-6
    for x, y in zip(pa, pb):
        sa, sb = oa.state[x], ob.state[y]
        assert set(sa.keys()) == set(sb.keys()) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sa["step"]) == float(sb["step"])
        assert rel_err(sb["exp_avg_sq"].cpu().numpy(), sa["exp_avg_sq"].cpu().numpy()) < 1e-6
    ob.load_state_dict(oa.state_dict())                      # interchangeable checkpoints
    pc, oc = make(GroupedAdam, weight_decay=0.1)
    for p in pc:
        p.grad = torch.ones_like(p)
    oc.step()                                                 # torch path, must not raise
    assert all(torch.isfinite(p).all() for p in pc)


def test_train_step_with_densification_on_gpu():
    """The reference's densify / prune / opacity-reset schedule inside train_step on the GPU: the Gaussian count changes,
    GroupedAdam's state follows the parameters, the fused mesh transform and the rasterizer pick up the new tensors, and
    training keeps working (finite loss, gradients on the re-created parameters at the next step)."""
    import bench_train as bt
    from types import SimpleNamespace
    from csplat import train as tr
    from gaussian_renderer import render
    from csplat.optim import GroupedAdam
    dev = torch.device("cuda")
    sc, pc, sim = bt.build(P=4000, W=160, H=128, grid=16, n_times=6, dev=dev)
    with torch.no_grad():
        pc._scaling.add_(0.9)
    bg = torch.ones(3, device=dev)
    times = [0.2, 0.4, 0.6]
    with torch.no_grad():
        keep = pc._features_dc.detach().clone()
        torch.manual_seed(0)
        pc._features_dc.add_(0.5 * torch.randn_like(pc._features_dc))
        targets_ = [render(c, pc, sim, tr.DEFAULT_PIPE, bg).render.clamp(0, 1).clone() for c in bt.cameras(sc, times, dev)]
        pc._features_dc.copy_(keep)
    with torch.no_grad():
        targets = targets_
    cams = bt.cameras(sc, times, dev, targets)
    pc.training_setup()
    pc.densification_setup(percent_dense=0.01)
    mopt = GroupedAdam(sim.parameters(), lr=3e-4)
    dopt = SimpleNamespace(densify_until_iter=100, densify_from_iter=2, densification_interval=3, opacity_reset_interval=9,
                           pruning_from_iter=2, pruning_interval=4, densify_grad_threshold_fine_init=2e-5,
                           densify_grad_threshold_after=2e-5, opacity_threshold_fine_init=0.05, opacity_threshold_fine_after=0.05,
                           cameras_extent=1.0, white_background=False, bary_cleanup=2)
    counts = []
    torch.manual_seed(0)
    for it in range(1, 10):
        p, l, stats = tr.train_step(it, cams, pc, sim, mopt, background=bg, densify_opt=dopt)
        assert torch.isfinite(l) and torch.isfinite(p)
        counts.append(pc.face_bary.shape[0])
        for grp in pc.optimizer.param_groups:
            q = grp["params"][0]
            assert q.shape[0] == counts[-1]
            st = pc.optimizer.state.get(q, {})
            assert not st or st["exp_avg"].shape == q.shape == st["exp_avg_sq"].shape     # (face_offset never gets a gradient)
        assert pc.face_ids.shape[0] == counts[-1] == pc.max_radii2D.shape[0] == pc.denom.shape[0]
    assert len(set(counts)) > 1 and max(counts) > 4000          # it did densify (and prune)


def test_capacity_store_kernels_and_reference_replay_on_gpu():
    """SURVEY 8(f) N3 on the device: (1) csplat_mask_to_map / csplat_rows_scatter against numpy on ragged row widths;
    (2) the reference's own densify / prune / opacity-reset run (tests/golden/densify.npz) replayed on GPU tensors through the
    capacity store: which rows survive, their order, the face ids, both Adam moments and the step counters as in the
    reference's run (values to fp32 rounding: the split's sampling arithmetic runs on another device than the fixture's), the
    SAME nn.Parameter objects before and after, and no allocation while the capacity suffices."""
    import ctypes as C
    import types
    from util import golden
    from csplat import native as n_, densify as dz
    from csplat.gaussians import MeshGaussians
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    # ---- (1) raw kernels
    for n in (1, 5, 1000, 70_001):
        mask = rng.random(n) < 0.37
        m8 = torch.tensor(mask.astype(np.uint8), device=dev)
        mp = torch.empty(n, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        tmp = torch.empty(int(n_.lib.csplat_mask_to_map_temp_bytes(n)), dtype=torch.uint8, device=dev)
        n_.check(n_.lib.csplat_mask_to_map(n_.stream_handle(dev), n, n_.ptr(m8), 11, n_.ptr(mp), n_.ptr(cnt), n_.ptr(tmp)), "map")
        want = np.where(mask, np.cumsum(mask) - 1 + 11, -1)
        np.testing.assert_array_equal(mp.cpu().numpy(), want)
        assert int(cnt.item()) == int(mask.sum())
        srcs = [torch.tensor(rng.normal(size=(n, w)).astype(np.float32), device=dev) for w in (1, 3, 45)] + \
               [torch.tensor(rng.integers(0, 1 << 40, n), device=dev)]
        dsts = [torch.full((n + 20,) + tuple(t.shape[1:]), 7, dtype=t.dtype, device=dev) for t in srcs] + [torch.full((n + 20, 4), 7.0, device=dev)]
        k = len(dsts)
        sp = (C.c_void_p * k)(*([t.data_ptr() for t in srcs] + [None]))
        dp = (C.c_void_p * k)(*[t.data_ptr() for t in dsts])
        rb = (C.c_int64 * k)(*[t[0].numel() * t.element_size() for t in dsts])
        n_.check(n_.lib.csplat_rows_scatter(n_.stream_handle(dev), k, C.cast(sp, C.c_void_p), C.cast(dp, C.c_void_p), C.cast(rb, C.c_void_p),
                                            n, n_.ptr(mp)), "scatter")
        for t, d in zip(srcs + [None], dsts):
            ref = np.full(tuple(d.shape), 7, dtype=d.cpu().numpy().dtype)
            ref[want[mask]] = 0 if t is None else t.cpu().numpy()[mask]
            np.testing.assert_array_equal(d.cpu().numpy(), ref)
    # ---- (2) the reference's run on the GPU
    g = golden("densify.npz")
    T = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    pc = MeshGaussians(3)
    pc.mesh = types.SimpleNamespace(pos=T(g["pos"]), face=T(g["face"]), edge_index=None)
    pc.face_ids = T(g["face_ids"])
    names = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    attrs = ["face_bary", "face_offset", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    for nm, a in zip(names, attrs):
        setattr(pc, a, torch.nn.Parameter(T(g["init." + nm])))
    pc.fused = False
    lrs = [1.6e-4, 1.6e-4, 2.5e-3, 2.5e-3 / 20, 0.05, 0.005, 0.001]
    from csplat.optim import GroupedAdam
    pc.optimizer = GroupedAdam([{"params": [getattr(pc, a)], "lr": lr, "name": nm} for a, lr, nm in zip(attrs, lrs, names)], lr=0.0, eps=1e-15)
    pc.densification_setup(percent_dense=0.01)
    pc.max_radii2D = T(g["max_radii2D"])
    objects = [getattr(pc, a) for a in attrs]

    def feed(flat):
        off = 0
        for grp in pc.optimizer.param_groups:
            p = grp["params"][0]
            p.grad = T(flat[off:off + p.numel()]).reshape(p.shape)
            off += p.numel()

    def check(tag, exact):
        for grp, obj in zip(pc.optimizer.param_groups, objects):
            p, nm = grp["params"][0], grp["name"]
            assert p is obj and p is getattr(pc, dict(zip(names, attrs))[nm])          # never re-created
            st = pc.optimizer.state[p]
            for got, key in ((p.detach(), f"{tag}.{nm}"), (st["exp_avg"], f"{tag}.{nm}.exp_avg"), (st["exp_avg_sq"], f"{tag}.{nm}.exp_avg_sq")):
                if exact:
                    np.testing.assert_array_equal(got.cpu().numpy(), g[key], err_msg=key)
                else:
                    np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-5, atol=1e-7, err_msg=key)
            assert float(st["step"]) == float(g[f"{tag}.{nm}.step"])
        np.testing.assert_array_equal(pc.face_ids.cpu().numpy(), g[f"{tag}.face_ids"])
        np.testing.assert_allclose(pc.pos_gradient_accum.cpu().numpy(), g[f"{tag}.pos_gradient_accum"], rtol=1e-6)
        np.testing.assert_array_equal(pc.denom.cpu().numpy(), g[f"{tag}.denom"])
        np.testing.assert_array_equal(pc.max_radii2D.cpu().numpy(), g[f"{tag}.max_radii2D"])

    flat, per = g["adam_grads"], sum(getattr(pc, a).numel() for a in attrs)
    for it in range(3):
        feed(flat[it * per:(it + 1) * per])
        pc.optimizer.step()
    check("stepped", exact=False)        # (GroupedAdam vs torch.optim.Adam on the CPU: 1e-6)
    vsp, upd = T(g["vsp"]), T(g["update_filter"])
    pc.add_densification_stats(vsp, upd)
    pc.add_densification_stats(vsp * 0.5, upd)
    real_normal = torch.normal
    try:     # the split draws from the global generator: replay the CPU stream the fixture was made with
        torch.manual_seed(4321)
        dz.torch.normal = lambda mean, std: real_normal(mean=mean.cpu(), std=std.cpu()).to(mean.device)
        pc.densify(2e-4, 0.05, 1.0, None)
    finally:
        dz.torch.normal = real_normal
    st_ = pc.store
    allocs = st_.allocations
    assert pc.face_bary.shape[0] == 81
    check("densified", exact=False)
    pc.prune(2e-4, 0.3, 1.0, 20)
    check("pruned", exact=False)
    pc.reset_opacity()
    check("reset", exact=False)
    feed(g["post_grads"])
    pc.optimizer.step()
    check("after_step", exact=False)
    assert pc.store is st_ and st_.allocations == allocs == 1                     # 60 -> 81 -> 65 rows inside capacity 1024
    assert pc.face_bary.data_ptr() == st_.sets[st_.live]["p:face_bary"].data_ptr()


@pytest.mark.parametrize("T,lams", [(3, (0.1, 0.3, 0.2)), (3, (0.0, 0.3, 0.0)), (3, (0.1, 0.0, 0.2)), (1, (0.1, 0.3, 0.2)),
                                    (2, (0.1, 0.3, 0.2)), (4, (0.1, 0.3, 0.2))])
def test_fused_cloth_regularisers_match_composed_torch(T, lams):
    """csplat_cloth_regs (deform-magnitude + rigidity + momentum, value and gradient in one launch) against the reference's
    composition from torch ops (train_utils.py:83-102) evaluated in fp64."""
    from types import SimpleNamespace
    from csplat import train as tr
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(17 + T)
    V, E = 3000, 17000
    D = torch.randn(T, V, 3, device=dev, generator=g)
    ei = torch.randint(0, V, (2, E), device=dev, generator=g)
    ei[1, :5] = ei[0, :5]                                     # zero-length edges: norm gradient 0 there
    if T >= 3:
        D[1, :7] = D[0, :7]                                   # zero deformation deltas
        D[1, 7:11] = D[0, 7:11]; D[2, 7:11] = D[0, 7:11]      # zero momentum (exactly, also in fp32: the sign of a rounding residue is not a parity question)
    rest = torch.rand(E, 1, device=dev, generator=g) + 0.5
    opt = SimpleNamespace(lambda_deform_mag=lams[0], lambda_rigid=lams[1], lambda_momentum=lams[2])
    pc = SimpleNamespace(mesh=SimpleNamespace(edge_index=ei), edge_norm=rest)
    Df = D.clone().requires_grad_()
    lf = tr.regularization(Df, pc, opt)
    assert type(lf.grad_fn).__name__.startswith("FusedClothRegs")
    (lf * 1.7).backward()
    D64 = D.double().requires_grad_()
    pc64 = SimpleNamespace(mesh=SimpleNamespace(edge_index=ei), edge_norm=rest.double())
    l64 = tr.regularization(D64, pc64, opt, fused=False)
    (l64 * 1.7).backward()
    assert abs(float(lf) - float(l64)) <= 2e-6 * max(abs(float(l64)), 1e-3), (float(lf), float(l64))
    if D64.grad is not None:
        err = float((Df.grad.double() - D64.grad).abs().max())
        assert err <= 2e-6 * max(float(D64.grad.abs().max()), 1e-6), err
    else:
        assert float(Df.grad.abs().max()) == 0.0
    # the loss value is summed in a fixed order; through the CSR of the graph the gradient is reproducible bit for bit too
    D2 = D.clone().requires_grad_()
    l2 = tr.regularization(D2, pc, opt)
    (l2 * 1.7).backward()
    assert float(l2) == float(lf) and torch.equal(D2.grad, Df.grad)
    # the scatter (atomics) form of the same kernel, used when no CSR is passed
    D3 = D.clone().requires_grad_()
    l3 = tr.FusedClothRegs.apply(D3, ei, rest.reshape(-1), *[max(x, 0.0) for x in lams], None)
    (l3 * 1.7).backward()
    assert abs(float(l3) - float(l64)) <= 2e-6 * max(abs(float(l64)), 1e-3)
    if D64.grad is not None:
        assert float((D3.grad.double() - D64.grad).abs().max()) <= 2e-6 * max(float(D64.grad.abs().max()), 1e-6)


@pytest.mark.parametrize("shape", [(3, 3, 64, 48), (1, 3, 37, 53)])
def test_fused_image_loss_matches_l1_plus_dssim(shape):
    """FusedImageLoss (csplat_l1 + csplat_ssim_fwd, one backward launch) == l1_loss + lambda (1 - ssim) composed from the
    two fused nodes, which the tests above pin against the reference formulas: value and gradient."""
    from types import SimpleNamespace
    from csplat import train as tr
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    img = torch.rand(*shape, device="cuda", generator=g)
    gt = (img + 0.2 * torch.randn(*shape, device="cuda", generator=g)).clamp(0, 1)
    opt = SimpleNamespace(lambda_dssim=0.2)
    a = img.clone().requires_grad_()
    la = tr.image_losses(a, gt, opt)
    assert type(la.grad_fn).__name__.startswith("FusedImageLoss")
    (la * 1.3).backward()
    b = img.clone().requires_grad_()
    lb = tr.l1_loss(b, gt) + opt.lambda_dssim * (1.0 - tr.ssim(b, gt))
    (lb * 1.3).backward()
    assert abs(float(la) - float(lb)) < 1e-6
    assert float((a.grad - b.grad).abs().max()) < 1e-6 * max(float(b.grad.abs().max()), 1e-12) + 1e-10


@pytest.mark.parametrize("masked", [False, True])
def test_fused_image_loss_side_outputs_and_weights(masked):
    """the one-launch step loss (csplat_image_loss_fwd / _bwd): out = w_img * (Ll1 + lambda dssim) + w_add * reg, the logged PSNR
    (utils/image_utils.py:17-21, unmasked, per camera) scaled, and the gradients towards the image AND the added scalar -- against the
    separately pinned nodes (FusedL1 / FusedSSIM / csplat_psnr) composed in torch.  Bit-reproducible from call to call."""
    from csplat import train as tr
    shape = (3, 3, 70, 90)
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand(*shape, device="cuda", generator=g)
    gt = (img + 0.1 * torch.randn(*shape, device="cuda", generator=g)).clamp(0, 1)
    mask = (torch.rand(shape[0], 1, *shape[2:], device="cuda", generator=g) > 0.3).float() if masked else None
    lam, w_img, w_add, ps = 0.05, 2.0 / 3.0, 0.5, 1.0 / 3.0
    a = img.clone().requires_grad_()
    reg = torch.tensor(0.37, device="cuda", requires_grad=True)
    loss, psnr, il = tr.FusedImageLoss.apply(a, gt, lam, mask, reg, w_img, w_add, ps)
    (loss * 1.7).backward()
    b = img.clone().requires_grad_()
    reg_b = torch.tensor(0.37, device="cuda", requires_grad=True)
    if masked:
        ssim_term = ((1.0 - tr.ssim(b, gt, return_map=True)) * mask).mean()
    else:
        ssim_term = 1.0 - tr.ssim(b, gt)
    il_b = tr.l1_loss(b, gt, mask) + lam * ssim_term
    want = w_img * il_b + w_add * reg_b
    (want * 1.7).backward()
    assert abs(float(loss) - float(want)) < 2e-6 and abs(float(il) - float(il_b)) < 2e-6
    assert abs(float(psnr) - ps * float(tr.psnr(img, gt).sum())) < 1e-4
    assert float((a.grad - b.grad).abs().max()) < 2e-6 * max(float(b.grad.abs().max()), 1e-12) + 1e-10
    assert abs(float(reg.grad) - float(reg_b.grad)) < 1e-6
    again = tr.FusedImageLoss.apply(img, gt, lam, mask, reg.detach(), w_img, w_add, ps)
    assert float(again[0]) == float(loss) and float(again[1]) == float(psnr)
    # a single [C, H, W] image (render() of one camera)
    one = tr.FusedImageLoss.apply(img[0], gt[0], lam)
    assert abs(float(one[0]) - float(tr.l1_loss(img[0], gt[0]) + lam * (1.0 - tr.ssim(img[0], gt[0])))) < 2e-6
    assert abs(float(one[1]) - float(tr.psnr(img[:1], gt[:1]))) < 1e-4


def test_step_stats_and_no_copy_unbind():
    """csplat_step_stats (train_utils.py:276-285 in one launch) against the torch composition, with a camera that received no
    gradient; UnbindViews' backward returns a VIEW when the incoming gradients sit back to back (the batched rasterizer's plan lays
    them out that way) and the stack otherwise; the regularisers' tap joins both gradients of the vertices in one tensor."""
    from types import SimpleNamespace
    from csplat import train as tr
    from csplat.gaussians import UnbindViews
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    P = 1237
    grads = [torch.randn(P, 3, device=dev, generator=g), None, torch.randn(P, 3, device=dev, generator=g)]
    radii = [torch.randint(0, 40, (P,), device=dev, dtype=torch.int32, generator=g) * (torch.rand(P, device=dev, generator=g) > 0.5).int()
             for _ in range(3)]
    vsg, rmax, vis = tr.step_stats(grads, radii, P, dev)
    assert torch.equal(vsg, grads[0] + grads[2])
    want = torch.stack(radii).max(0).values
    assert torch.equal(rmax, want) and torch.equal(vis, want > 0) and vis.dtype == torch.bool
    # --- UnbindViews
    x = torch.randn(3, P, 4, device=dev, generator=g, requires_grad=True)
    ys = UnbindViews.apply(x)
    assert all(torch.equal(a, b) for a, b in zip(ys, x.unbind(0)))
    buf = torch.randn(3 * P * 4 + 64, device=dev, generator=g)
    gs = [buf[i * P * 4:(i + 1) * P * 4].view(P, 4) for i in range(3)]
    (gx,) = torch.autograd.grad(ys, x, gs, retain_graph=True)
    assert gx.data_ptr() == buf.data_ptr() and torch.equal(gx, torch.stack(gs))            # a view, no copy
    gs2 = [gs[0], gs[2].clone()]                      # not back to back, and the third row receives no gradient at all
    (gx2,) = torch.autograd.grad(ys[:2], x, gs2)
    assert torch.equal(gx2, torch.stack([gs2[0], gs2[1], torch.zeros_like(gs2[0])]))
    # --- the regulariser tap: d(reg + f(vertices)) / d vertices in one tensor == the two-path autograd sum
    T, V = 3, 400
    D = (0.1 * torch.randn(T, V, 3, device=dev, generator=g)).requires_grad_()
    ei = torch.stack([torch.arange(V - 1, device=dev), torch.arange(1, V, device=dev)])
    pc = SimpleNamespace(mesh=SimpleNamespace(edge_index=ei), edge_norm=torch.rand(V - 1, 1, device=dev, generator=g) * 0.1)
    w = torch.randn(T, V, 3, device=dev, generator=g)
    reg, Dt = tr.regularization(D, pc, tr.DEFAULT_OPT, tap=True)
    (1.3 * reg + (Dt * w).sum()).backward()
    D2 = D.detach().clone().requires_grad_()
    (1.3 * tr.regularization(D2, pc, tr.DEFAULT_OPT) + (D2 * w).sum()).backward()
    assert float((D.grad - D2.grad).abs().max()) <= 1e-6 * float(D2.grad.abs().max())


def test_train_step_static_stage_on_gpu():
    """train_step(static=True) -- the reference's static stage (train_utils.py:240-321 with render_static: the rest mesh, no simulator,
    no regularisers): the Gaussian parameters train, the simulator's stay untouched, batched and camera-by-camera agree."""
    import bench_train as bt
    from csplat import train as tr
    from gaussian_renderer import render
    dev = torch.device("cuda")

    def setup():
        torch.manual_seed(4)
        sc, pc, sim = bt.build(P=1500, W=80, H=64, grid=10, n_times=5, dev=dev)
        with torch.no_grad():
            pc._scaling.add_(0.8)
        bg = torch.ones(3, device=dev)
        cams = bt.cameras(sc, [0.25, 0.5, 0.75], dev)
        with torch.no_grad():
            keep = pc._features_dc.detach().clone()
            torch.manual_seed(0)
            pc._features_dc.add_(0.5 * torch.randn_like(pc._features_dc))
            targets = [render(c, pc, sim, tr.DEFAULT_PIPE, bg, render_static=True).render.clamp(0, 1).clone() for c in cams]
            pc._features_dc.copy_(keep)
        cams = bt.cameras(sc, [0.25, 0.5, 0.75], dev, targets)
        pc.training_setup(feature_lr=0.02)
        mopt = torch.optim.Adam(sim.parameters(), lr=3e-4)
        return pc, sim, mopt, cams, bg
    runs = []
    for batched in (True, False):
        pc, sim, mopt, cams, bg = setup()
        before = [p.detach().clone() for p in sim.parameters()]
        ps = []
        for it in range(1, 9):
            p_, loss, stats = tr.train_step(it, cams, pc, sim, mopt, background=bg, static=True, batched_views=batched)
            ps.append(float(p_))
        assert ps[-1] > ps[0] + 0.2, ps
        for a, b in zip(before, sim.parameters()):
            assert torch.equal(a, b.detach())
        assert stats["radii"].shape == (1500,) and stats["viewspace_grad"].shape == (1500, 3)
        runs.append(ps)
    assert max(abs(a - b) for a, b in zip(*runs)) < 2e-3, runs


def test_step_head_nodes_equal_the_chained_nodes():
    """SimulatorStep (simulator + regularisers) and GaussianStepInputs (mesh transform + activations) -- the two autograd nodes the
    batched train step starts with -- against the nodes they replace (forward_times + regularization(tap), transform_views +
    activations): the same launches, hence the same bits forward and the same parameter gradients."""
    import bench_train as bt
    from csplat import train as tr
    dev = torch.device("cuda")
    torch.manual_seed(3)
    sc, pc, sim = bt.build(P=2000, W=64, H=64, grid=12, n_times=6, dev=dev)
    with torch.no_grad():
        sim.output.weight.mul_(300.0)
    times = [0.2, 0.4, 0.6]
    g = torch.Generator(device="cuda").manual_seed(1)
    params = list(sim.parameters()) + list(pc.parameters())

    def grads():
        out = [None if p.grad is None else p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        return out
    # chained
    D1 = sim.forward_times(times)
    reg1, D1t = tr.regularization(D1, pc, tr.DEFAULT_OPT, tap=True)
    xyz1, quat1 = pc.transform_views(D1t)
    op1, sc1, sh1 = pc.activations()
    w = [torch.randn(t.shape, device=dev, generator=g) for t in (xyz1[0], quat1[0], op1, sc1, sh1)]
    val = lambda xyz, quat, op, sc_, sh, reg: (sum((x * w[0]).sum() for x in xyz) + sum((q * w[1]).sum() for q in quat) + (op * w[2]).sum() +  # noqa: E731
                                               (sc_ * w[3]).sum() + (sh * w[4]).sum() + 1.7 * reg)
    val(xyz1, quat1, op1, sc1, sh1, reg1).backward()
    g1 = grads()
    # one node each
    head = tr.simulator_step(sim, times, pc, tr.DEFAULT_OPT)
    assert head is not None and type(head[0].grad_fn).__name__.startswith("SimulatorStep")
    D2, reg2 = head
    both = pc.step_inputs(D2)
    assert both is not None and type(both[2].grad_fn).__name__.startswith("GaussianStepInputs")
    assert torch.equal(D2, D1) and torch.equal(reg2, reg1)
    for a, b in zip(list(both[0]) + list(both[1]) + list(both[2:]), list(xyz1) + list(quat1) + [op1, sc1, sh1]):
        assert torch.equal(a, b)
    val(both[0], both[1], both[2], both[3], both[4], reg2).backward()
    g2 = grads()
    for a, b, p in zip(g1, g2, params):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1e-20), tuple(p.shape)
    # deferred form: nothing is computed until launch_deferred()
    D3, reg3 = tr.simulator_step(sim, times, pc, tr.DEFAULT_OPT, defer=True)
    tr.launch_deferred()
    assert torch.equal(reg3, reg1)


def test_train_step_camera_by_camera_equals_batched():
    """train_step(batched_views=False) -- render() per camera, simulator per camera, composed losses as upstream's loop --
    against the default batched step (render_views, forward_times, fused nodes): same PSNR, loss, statistics and the same
    parameters after one optimisation step from the same state."""
    import bench_train as bt
    from csplat import train as tr
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    times = [0.2, 0.4, 0.6]
    res = []
    for batched in (True, False):
        torch.manual_seed(123)            # (the simulator's input / hidden layers are randomly initialised)
        sc, pc, sim = bt.build(P=3000, W=112, H=96, grid=14, n_times=6, dev=dev)
        with torch.no_grad():
            pc._scaling.add_(0.9)
            torch.manual_seed(1)
            sim.output.weight.copy_(1e-3 * torch.randn_like(sim.output.weight))
        bg = torch.ones(3, device=dev)
        with torch.no_grad():
            keep = pc._features_dc.detach().clone()
            torch.manual_seed(0)
            pc._features_dc.add_(0.5 * torch.randn_like(pc._features_dc))
            targets = [render(c, pc, sim, tr.DEFAULT_PIPE, bg).render.clamp(0, 1).clone() for c in bt.cameras(sc, times, dev)]
            pc._features_dc.copy_(keep)
        cams = bt.cameras(sc, times, dev, targets)
        pc.training_setup(feature_lr=0.01)
        mopt = torch.optim.Adam(sim.parameters(), lr=3e-4)
        p, l, stats = tr.train_step(1, cams, pc, sim, mopt, background=bg, batched_views=batched)
        res.append((float(p), float(l), stats, [q.detach().clone() for q in pc.parameters()],
                    [q.detach().clone() for q in sim.parameters()]))
    a, b = res
    assert abs(a[0] - b[0]) < 1e-3 and abs(a[1] - b[1]) < 1e-6 * max(abs(b[1]), 1.0), (a[:2], b[:2])
    assert torch.equal(a[2]["radii"], b[2]["radii"]) and torch.equal(a[2]["visibility_filter"], b[2]["visibility_filter"])
    assert rel_err(a[2]["viewspace_grad"].cpu().numpy(), b[2]["viewspace_grad"].cpu().numpy()) < 1e-4
    # parameters after the Adam step: the first step moves every entry by ~lr * sign(grad); entries whose gradient is at
    # rounding level can flip sign between the two summation orders, so compare against the step size
    for qa, qb in zip(a[3] + a[4], b[3] + b[4]):
        d = (qa - qb).abs()
        assert float((d > 1e-6 + 1e-3 * qb.abs()).float().mean()) < 0.02, float(d.max())


def test_fused_gaussian_activations_match_the_properties():
    """MeshGaussians.activations() (csplat_gauss_act_fwd / _bwd: one launch each way) == (get_opacity, get_scaling, get_features) of
    gaussian_model.py:96-121, values to 1 ulp-level and gradients to 1e-6 relative"""
    from csplat.gaussians import _GaussianActivations
    gen = torch.Generator().manual_seed(4)
    P = 1237
    raw = [torch.randn(P, 1, generator=gen) * 3, torch.randn(P, 3, generator=gen), torch.randn(P, 1, 3, generator=gen),
           torch.randn(P, 15, 3, generator=gen)]
    a = [t.clone().cuda().requires_grad_() for t in raw]
    b = [t.clone().cuda().requires_grad_() for t in raw]
    w = [torch.randn(P, 1, generator=gen).cuda(), torch.randn(P, 3, generator=gen).cuda(), torch.randn(P, 16, 3, generator=gen).cuda()]
    out = _GaussianActivations.apply(*a)
    ref = (torch.sigmoid(b[0]), torch.exp(b[1]), torch.cat((b[2], b[3]), dim=1))
    for o, r in zip(out, ref):
        assert o.shape == r.shape and float((o - r).abs().max()) <= 2e-7 * float(r.abs().max())
    sum((o * wi).sum() for o, wi in zip(out, w)).backward()
    sum((r * wi).sum() for r, wi in zip(ref, w)).backward()
    for x, y in zip(a, b):
        assert float((x.grad - y.grad).abs().max()) <= 1e-6 * float(y.grad.abs().max())
    # an output nothing depends on: its incoming gradient is None inside the node
    c = [t.clone().cuda().requires_grad_() for t in raw]
    o2 = _GaussianActivations.apply(*c)
    (o2[1] * w[1]).sum().backward()
    assert float(c[0].grad.abs().max()) == 0.0 and float(c[3].grad.abs().max()) == 0.0
    assert float((c[1].grad - b[1].grad).abs().max()) <= 1e-6 * float(b[1].grad.abs().max())



def _captured_fixture(seed=3):
    import bench_train as bt
    from csplat import train as tr
    from csplat.optim import GroupedAdam
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    sc, pc, sim = bt.build(P=4000, W=160, H=128, grid=16, n_times=6, dev=dev)
    with torch.no_grad():
        pc._scaling.add_(0.9)
        sim.output.weight.copy_(1e-3 * torch.randn_like(sim.output.weight))
    bg = torch.ones(3, device=dev)
    times = [0.2, 0.4, 0.6]
    with torch.no_grad():
        keep = pc._features_dc.detach().clone()
        pc._features_dc.add_(0.5 * torch.randn_like(pc._features_dc))
        targets = [render(c, pc, sim, tr.DEFAULT_PIPE, bg).render.clamp(0, 1).clone() for c in bt.cameras(sc, times, dev)]
        pc._features_dc.copy_(keep)
    cams = bt.cameras(sc, times, dev, targets)
    pc.training_setup(feature_lr=0.01)
    mopt = GroupedAdam(sim.parameters(), lr=3e-4)
    return pc, sim, mopt, cams, bg


DET = 256       # csplat_debug_flags bit 8: bit-reproducible K7 (per-(entry, block) records summed in emission order instead of float atomics)


@contextlib.contextmanager
def _reproducible_k7(on=True):
    """Both arms of an eager-against-recorded comparison run with the bit-reproducible K7: every kernel of the step then sums in a fixed
    order, and a recorded step that does the eager step's work must reproduce it BIT FOR BIT -- a statement no tolerance can water down and
    no threshold tie can flake.  The flag is set before any scratch of the step is sized and reset whatever happens."""
    from csplat import native
    native.lib.csplat_debug_flags(DET if on else 0)
    try:
        yield
    finally:
        native.lib.csplat_debug_flags(0)


def _adam_state(opt, params):
    out = []
    for p in params:
        st = opt.state.get(p)
        if st:
            out += [st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
    return out


def _captured_run(mode, iterations, cams_of, seed=3, sh_at_max=False, before_step=None, det=True):
    """one training run from `_captured_fixture(seed)`: per step (PSNR, loss, radii, viewspace gradient sum, visibility), then every
    parameter, both Adam moments of both optimizers, the host and device step counts and the CapturedStep object"""
    from csplat import train as tr
    with _reproducible_k7(det):
        pc, sim, mopt, cams, bg = _captured_fixture(seed=seed)
        if sh_at_max:
            pc.active_sh_degree = pc.max_sh_degree
        log = []
        for it in iterations:
            if before_step is not None:
                before_step(it, pc, sim, cams, mode)
            ps, loss, stats = tr.train_step(it, cams_of(it, cams), pc, sim, mopt, background=bg, captured=(mode == "captured"))
            log.append((float(ps), float(loss), stats["radii"].clone(), stats["viewspace_grad"].clone(), stats["visibility_filter"].clone()))
        torch.cuda.synchronize()
        params = [p.detach().clone() for p in list(pc.parameters()) + list(sim.parameters())]
        moments = _adam_state(pc.optimizer, pc.parameters()) + _adam_state(mopt, sim.parameters())
        steps = [float(pc.optimizer.state[p]["step"]) for p in pc.parameters() if pc.optimizer.state.get(p)] + \
                [float(mopt.state[p]["step"]) for p in sim.parameters()]
        dev_steps = None
        if mode == "captured":
            dev_steps = (int(pc.optimizer._cap["state"].item()), int(mopt._cap["state"].item()))
    return {"log": log, "params": params, "moments": moments, "steps": steps, "dev_steps": dev_steps,
            "cs": getattr(pc, "_captured_step", None)}


def _assert_runs_bit_equal(a, b):
    """two runs in the bit-reproducible mode: PSNR, loss, radii, visibility and the summed screen-space gradient of EVERY step, every
    parameter, both Adam moments and the step counts -- equal bit for bit"""
    assert len(a["log"]) == len(b["log"])
    for k_, ((pa, la, ra, va, fa), (pb, lb, rb, vb, fb)) in enumerate(zip(a["log"], b["log"])):
        assert pa == pb and la == lb, (k_, pa, pb, la, lb)
        assert torch.equal(ra, rb) and torch.equal(fa, fb), k_
        assert torch.equal(va, vb), (k_, float((va - vb).abs().max()))
    assert len(a["params"]) == len(b["params"]) and len(a["moments"]) == len(b["moments"]) and len(a["moments"]) > 0
    for i, (x, y) in enumerate(zip(a["params"], b["params"])):
        assert torch.equal(x, y), ("parameter", i, tuple(x.shape), float((x - y).abs().max()))
    for i, (x, y) in enumerate(zip(a["moments"], b["moments"])):
        assert torch.equal(x, y), ("moment", i, tuple(x.shape), float((x - y).abs().max()))
    assert a["steps"] == b["steps"]


def test_captured_train_step_equals_the_eager_step():
    """csplat.train.CapturedStep (train_step(captured=True)): the step recorded once into a hipGraph -- forward launched on faith
    (csplat_forward_views_faith), Adam's step count / learning rates / go word on the device (csplat_adam_step_dev), ONE host read at the
    end -- against the eager train_step from the same initial state, eight steps, BOTH in the bit-reproducible mode: PSNR, loss, the
    densification statistics of every step, every parameter, both Adam moments and the step counters equal BIT FOR BIT.
    Reference step: scene_reconstruction/train_utils.py:240-321, timed as train.py:146,178."""
    same = lambda it, cams: cams  # noqa: E731
    its = list(range(1, 9))
    eager = _captured_run("eager", its, same)
    cap = _captured_run("captured", its, same)
    cs = cap["cs"]
    assert cs is not None and cs.stats["recorded"] == 1 and cs.stats["replayed"] == 7 and cs.stats["eager"] == 1 and cs.stats["missed"] == 0, cs.stats
    assert eager["steps"] == cap["steps"] == [8.0] * len(eager["steps"])
    assert cap["dev_steps"] == (8, 8)
    _assert_runs_bit_equal(eager, cap)
    # and the reproducible mode is reproducible: a second eager run equals the first
    _assert_runs_bit_equal(eager, _captured_run("eager", its, same))


def _calibration():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "captured_atomic_calibration.json")
    with open(path) as f:
        return json.load(f)


def captured_atomic_metrics(eager, cap):
    """How far two runs in the DEFAULT mode (K7 sums with float atomics: the order of the additions differs from run to run) are apart.
    tools/calibrate_captured_atomic.py measures these over many pairs of runs; the test's bars are twice the largest values it saw."""
    m = {"psnr": 0.0, "loss_rel_first": 0.0, "loss_rel": 0.0, "radii_frac": 0.0, "vgrad_rel": 0.0, "param_frac": 0.0}
    for k_, ((pa, la, ra, va, fa), (pb, lb, rb, vb, fb)) in enumerate(zip(eager["log"], cap["log"])):
        m["psnr"] = max(m["psnr"], abs(pa - pb))
        rel = abs(la - lb) / max(abs(la), 1e-3)
        if k_ == 0:
            m["loss_rel_first"] = rel
        m["loss_rel"] = max(m["loss_rel"], rel)
        m["radii_frac"] = max(m["radii_frac"], float((ra != rb).float().mean()))
        m["vgrad_rel"] = max(m["vgrad_rel"], float((va - vb).abs().max()) / max(float(va.abs().max()), 1e-20))
    for x, y in zip(eager["params"] + eager["moments"], cap["params"] + cap["moments"]):
        d = (x - y).abs()
        m["param_frac"] = max(m["param_frac"], float((d > 1e-6 + 2e-3 * y.abs()).float().mean()))
    return m


def test_captured_train_step_default_mode_within_calibrated_atomic_noise():
    """The same comparison in the DEFAULT mode, where K7 adds with float atomics and two runs are two samples of a chaotic trajectory (one
    compositing threshold decided the other way at one pixel moves a loss by ~1e-4 of itself).  The bars are not tuned by hand: they are
    twice the largest deviation over the committed calibration (tests/golden/captured_atomic_calibration.json, made by
    tools/calibrate_captured_atomic.py from >= 32 pairs of runs, eager against eager AND eager against captured; distribution in
    profiles/r06_captured_atomic_calibration.txt)."""
    cal = _calibration()
    assert cal["pairs"] >= 32
    same = lambda it, cams: cams  # noqa: E731
    its = list(range(1, 9))
    # (seed 11: of the calibration's four scenes the one whose pairs stayed below 2e-4 in every metric -- seed 3 sits one threshold tie
    #  (vgrad 1.6e-2, PSNR 1.1e-3) below its bars in half of its pairs, and a second tie in one run would cross the PSNR bar)
    eager = _captured_run("eager", its, same, seed=11, det=False)
    cap = _captured_run("captured", its, same, seed=11, det=False)
    cs = cap["cs"]
    assert cs.stats["recorded"] == 1 and cs.stats["replayed"] == 7 and cs.stats["missed"] == 0, cs.stats
    assert eager["steps"] == cap["steps"] == [8.0] * len(eager["steps"])
    m = captured_atomic_metrics(eager, cap)
    for k, v in m.items():
        assert v <= cal["bars"][k], (k, v, cal["bars"][k], m)
    for (_p, _l, _r, _v, fc), (_p2, _l2, rc, _v2, _f2) in zip(cap["log"], cap["log"]):
        assert torch.equal(fc, rc > 0)


def test_captured_train_step_survives_a_miss():
    """The counts outgrow the capacities a graph was recorded with (every Gaussian grows by e^0.6 between two steps: R up ~2x): the
    replay's `valid` word comes back 0 and NOTHING was applied -- parameters, moments and step counts bit-identical to before the
    replay -- the step is repeated eagerly, the graph re-recorded, and training goes on; the run equals, bit for bit (reproducible mode), an
    all-eager run that takes the same jump.  (the forced-miss test.)"""
    def jump(it, pc, sim, cams, mode):
        if it != 5:
            return
        with torch.no_grad():
            pc._scaling.add_(0.6)
        if mode == "captured":      # what a replay that misses must leave untouched
            before = [p.detach().clone() for p in list(pc.parameters()) + list(sim.parameters())]
            mom = _adam_state(pc.optimizer, pc.parameters())
            cs = pc._captured_step
            key = cs._key(cams)
            st = cs.graphs[key]
            cs._fill(st, cams)
            st["graph"].replay()
            torch.cuda.synchronize()
            assert float(st["host"][1]) == 0.0, "the jump was meant to overflow the recorded capacities"
            for a, b in zip(before, list(pc.parameters()) + list(sim.parameters())):
                assert torch.equal(a, b)
            for a, b in zip(mom, _adam_state(pc.optimizer, pc.parameters())):
                assert torch.equal(a, b)
            assert [float(pc.optimizer.state[p]["step"]) for p in pc.parameters() if pc.optimizer.state.get(p)] == [4.0] * 6
            assert int(pc.optimizer._cap["state"].item()) == 4
    same = lambda it, cams: cams  # noqa: E731
    its = list(range(1, 9))
    eager = _captured_run("eager", its, same, seed=5, before_step=jump)
    cap = _captured_run("captured", its, same, seed=5, before_step=jump)
    cs = cap["cs"]
    assert cs.stats["missed"] == 1 and cs.stats["recorded"] == 2, cs.stats
    _assert_runs_bit_equal(eager, cap)


def test_captured_train_step_across_an_eager_iteration():
    """CapturedStep runs every `iteration % 1000 == 0` step through the eager train_step (the step that may raise the SH
    degree, train_utils.py:249-251).  With the degree already at its maximum the step shape -- and so the recorded graph -- survives that
    step; the eager step advances the optimizers' HOST step counters only, and the next replay used to read a stale device count (wrong
    bias correction, then a sequence mismatch and RuntimeError).  Iterations 997..1004, captured against eager in the reproducible mode:
    bit-equal losses, parameters and moments, host and device step counts equal at the end, the graph recorded once."""
    its = list(range(997, 1005))
    same = lambda it, cams: cams  # noqa: E731
    eager = _captured_run("eager", its, same, sh_at_max=True)
    cap = _captured_run("captured", its, same, sh_at_max=True)
    cs = cap["cs"]
    # 997 eager (first of its shape), 998-999 record + replay, 1000 eager, 1001-1004 replays of the SAME graph
    assert cs.stats["recorded"] == 1 and cs.stats["eager"] == 2 and cs.stats["replayed"] == 6 and cs.stats["missed"] == 0, cs.stats
    assert eager["steps"] == cap["steps"] == [8.0] * len(eager["steps"])
    assert cap["dev_steps"] == (8, 8)
    _assert_runs_bit_equal(eager, cap)


def test_captured_train_step_alternating_step_shapes():
    """one graph per step SHAPE.  Recording a second shape used to re-allocate the optimizers' device-side step
    count and learning-rate table, leaving the first shape's graph with dangling pointers.  Three cameras and two cameras alternate
    (both shapes recorded, then each replayed after the other was recorded); the learning rate of one group changes mid-run (a
    schedule editing param_groups, utils/general_utils.py:32-65 as used by gaussian_model.py:162-168): captured == eager, bit for bit."""
    its = list(range(1, 13))
    pick = lambda it, cams: cams if (it // 2) % 2 == 0 else cams[:2]  # noqa: E731

    def sched(it, pc, sim, cams, mode):
        if it == 9:
            for g in pc.optimizer.param_groups:
                g["lr"] = g["lr"] * 0.5
    eager = _captured_run("eager", its, pick, seed=7, before_step=sched)
    cap = _captured_run("captured", its, pick, seed=7, before_step=sched)
    cs = cap["cs"]
    assert cs.stats["recorded"] == 2 and cs.stats["missed"] == 0 and cs.stats["replayed"] >= 8, cs.stats
    assert eager["steps"] == cap["steps"] == [12.0] * len(eager["steps"])
    assert cap["dev_steps"] == (12, 12)
    _assert_runs_bit_equal(eager, cap)


def test_captured_train_step_rerecords_after_a_scratch_eviction():
    """a failed entry point (csplat.native.check) or an overflowing cache drops the "zeroed once" scratch buffers the
    recorded graphs point into.  The eviction bumps csplat.native.SCRATCH_EPOCH; CapturedStep must not replay a graph recorded under an
    older epoch -- it records again -- and the run still equals the eager run.  Run in the DEFAULT mode as well as the reproducible one: the
    persistent zeroed records (the cache the eviction drops) exist only in the default mode."""
    from csplat import native
    its = list(range(1, 9))
    same = lambda it, cams: cams  # noqa: E731

    def evict(it, pc, sim, cams, mode):
        if it == 5 and getattr(pc, "_captured_step", None) is not None:
            native.evict_scratch()
    eager = _captured_run("eager", its, same, seed=11)
    cap = _captured_run("captured", its, same, seed=11, before_step=evict)
    cs = cap["cs"]
    assert cs.stats["recorded"] == 2 and cs.stats.get("rerecorded_stale") == 1 and cs.stats["missed"] == 0, cs.stats
    _assert_runs_bit_equal(eager, cap)
    # default mode: the recording really points into the evicted cache; bars from the committed calibration
    cal = _calibration()
    eager = _captured_run("eager", its, same, seed=11, det=False)
    cap = _captured_run("captured", its, same, seed=11, before_step=evict, det=False)
    cs = cap["cs"]
    assert cs.stats["recorded"] == 2 and cs.stats.get("rerecorded_stale") == 1 and cs.stats["missed"] == 0, cs.stats
    m = captured_atomic_metrics(eager, cap)
    for k, v in m.items():
        assert v <= cal["bars"][k], (k, v, cal["bars"][k], m)
