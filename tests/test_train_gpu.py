"""train_step analogue on a small scene: the whole hot path under autograd + two Adam
optimisers must fit a perturbed target: PSNR rises, loss falls, statistics have the reference's shapes."""
import pytest

import util  # noqa: F401

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
