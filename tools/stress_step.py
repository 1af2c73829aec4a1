#!/usr/bin/env python3
"""stress_step.py [n]: ONE train step from the same state n times; every by-product is compared with the first run
(images bit for bit, gradients to 1e-5 of their scale) to localise rare glitches."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench_train as bt
from csplat import synthetic as syn, train as tr
from csplat.gaussians import MeshGaussians
from gaussian_renderer import render, render_views
from meshnet.meshnet_network import ResidualMeshSimulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P, W, H, grid, n_times = 400, 48, 48, 8, 4
if len(sys.argv) > 2:
    P, W, H, grid = 20000, 400, 400, 40
sc = syn.scene_1(P=P, W=W, H=H, n_cams=1, grid=grid, n_times=n_times, seed=77)
if P == 400:
    sc["log_scales"] = sc["log_scales"] + np.log(6.0)
times = [1 / 3, 2 / 3, 1.0]
dev = torch.device("cuda:0")
T = lambda a, d=torch.float32: torch.tensor(a, device=dev, dtype=d)  # noqa: E731
pc = MeshGaussians(3).from_arrays(T(sc["mesh_pos"][0]), T(sc["faces"].T.copy(), torch.long), T(sc["edge_index"], torch.long),
                                  T(sc["face_ids"], torch.long), T(sc["bary"]), T(sc["log_scales"]), T(sc["quats"]),
                                  T(sc["opacity_logits"]), T(sc["sh"]))
pc.active_sh_degree = 3
sim = ResidualMeshSimulator(T(sc["mesh_pos"]), device=dev)
torch.manual_seed(5)
with torch.no_grad():
    sim.output.weight.copy_(torch.randn(sc["mesh_pos"].shape[1] * 3, 256, device=dev) * 1e-3)
bg = torch.ones(3, device=dev)
with torch.no_grad():
    keep = [p.detach().clone() for p in pc.parameters()]
    torch.manual_seed(9)
    pc._features_dc.add_(0.6 * torch.randn(P, 1, 3, device=dev))
    pc._opacity.add_(0.8 * torch.randn(P, 1, device=dev))
    targets = [render(c, pc, sim, tr.DEFAULT_PIPE, bg).render.clamp(0, 1).clone() for c in bt.cameras(sc, times, dev)]
    for p, k in zip(pc.parameters(), keep):
        p.copy_(k)
cams = bt.cameras(sc, times, dev, targets)
gt = torch.stack(targets)
params = list(pc.parameters()) + list(sim.parameters())
names = [f"pc{i}{tuple(p.shape)}" for i, p in enumerate(pc.parameters())] + [f"sim{i}{tuple(p.shape)}" for i, p in enumerate(sim.parameters())]


def one():
    for p in params:
        p.grad = None
    deforms = sim.forward_times([c.time for c in cams])
    reg = tr.regularization(deforms, pc, tr.DEFAULT_OPT)
    pkgs, stacked = render_views(cams, pc, sim, tr.DEFAULT_PIPE, bg, return_stacked=True, vertice_deforms=deforms)
    ps = tr.psnr(stacked, gt)
    loss = tr.image_losses(stacked, gt, tr.DEFAULT_OPT) + reg
    loss.backward()
    out = {"deforms": deforms.detach().clone(), "reg": reg.detach().clone(), "image": stacked.detach().clone(), "psnr": ps.clone(),
           "loss": loss.detach().clone(), "radii": torch.stack([p.radii for p in pkgs]).clone(),
           "vsp": torch.stack([p.viewspace_points.grad for p in pkgs]).clone()}
    for nm, p in zip(names, params):
        out["g:" + nm] = None if p.grad is None else p.grad.detach().clone()
    return out


ref = one()
exact = ("deforms", "image", "psnr", "radii", "reg")
bad = 0
for r in range(n):
    cur = one()
    for k, v in cur.items():
        if v is None:
            continue
        a, b = v.double(), ref[k].double()
        if k in exact:
            ok = torch.equal(v, ref[k])
            e = float((a - b).abs().max())
        else:
            e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            ok = e < 1e-5
        if not ok:
            bad += 1
            print(f"run {r}: {k} deviates: {e:.3e}")
torch.cuda.synchronize()
print(f"{n} runs, {bad} deviations")
