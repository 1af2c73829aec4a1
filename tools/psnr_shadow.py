"""tools/psnr_shadow.py [steps] [lo] [hi] -- GRADIENT parity ALONG the HIP training trajectory of tests/test_psnr_parity_gpu.py: after
every HIP step in [lo, hi] the current parameters are copied to the CPU and the full step loss (simulator + mesh transform + rasterizer +
L1 + 0.05 (1 - SSIM) + cloth regularisers) is differentiated twice -- on the GPU through the HIP path and on the CPU through the fp64
oracle -- and the gradients compared per parameter.  Answers: is a dip of the HIP trajectory the dynamics of fp32 training, or a kernel
that goes wrong on some state?  GPU box."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import test_psnr_parity_gpu as t
from csplat import train as tr
from gaussian_renderer import render

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 460
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 415
hi = int(sys.argv[3]) if len(sys.argv) > 3 else steps
state = {}


def loss_of(pc, sim, cams, bg, cpu):
    imgs, verts = [], []
    for c in cams:
        if cpu:
            color, v = t._oracle_render(c, pc, sim, np.ones(3))
        else:
            r = render(c, pc, sim, tr.DEFAULT_PIPE, bg)
            color, v = r.render, r.vertice_deform
        imgs.append(color.unsqueeze(0)); verts.append(v[None])
    image = torch.cat(imgs, 0)
    gt = torch.stack([c.original_image for c in cams])
    return tr.image_losses(image, gt, tr.DEFAULT_OPT) + tr.regularization(torch.cat(verts, 0), pc, tr.DEFAULT_OPT), image, gt


def hook(it, pc, sim, cams, bg, build, psnr):
    if not (lo <= it <= hi):
        return
    if "pc_c" not in state:
        import bench_train as bt
        state["pc_c"], state["sim_c"] = build("cpu", torch.float64)
        state["pc_c"].fused = False
        state["cams_c"] = []
        for c in cams:
            cc = type(c)(**{k: (v.detach().cpu().double() if torch.is_tensor(v) else v) for k, v in vars(c).items()})
            state["cams_c"].append(cc)
    pc_c, sim_c = state["pc_c"], state["sim_c"]
    with torch.no_grad():
        for a, b in zip(pc_c.parameters(), pc.parameters()):
            a.copy_(b.detach().cpu().double())
        for a, b in zip(sim_c.parameters(), sim.parameters()):
            a.copy_(b.detach().cpu().double())
    for p in list(pc.parameters()) + list(sim.parameters()) + list(pc_c.parameters()) + list(sim_c.parameters()):
        p.grad = None
    lg, img_g, _ = loss_of(pc, sim, cams, bg, False)
    lg.backward()
    lc, img_c, _ = loss_of(pc_c, sim_c, state["cams_c"], None, True)
    lc.backward()
    # third evaluation: the SAME fp64 torch graph around the fp32 build of the C oracle's rasterizer
    g64 = [None if p.grad is None else p.grad.clone() for p in list(pc_c.parameters()) + list(sim_c.parameters())]
    for p in list(pc_c.parameters()) + list(sim_c.parameters()):
        p.grad = None
    t.ORACLE_DTYPE[0] = np.float32
    l32, _, _ = loss_of(pc_c, sim_c, state["cams_c"], None, True)
    l32.backward()
    t.ORACLE_DTYPE[0] = np.float64
    g32 = [None if p.grad is None else p.grad.clone() for p in list(pc_c.parameters()) + list(sim_c.parameters())]
    for p, gg in zip(list(pc_c.parameters()) + list(sim_c.parameters()), g64):
        p.grad = gg
    names = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + [n for n, _ in sim.named_parameters()]
    errs = []
    for n, a, b in zip(names, list(pc.parameters()) + list(sim.parameters()), list(pc_c.parameters()) + list(sim_c.parameters())):
        if a.grad is None or b.grad is None:
            continue
        ga, gb = a.grad.detach().cpu().double(), b.grad
        errs.append((n, float((ga - gb).abs().max() / (gb.abs().max() + 1e-30))))
    e32 = []
    for n, a, b, c in zip(names, list(pc.parameters()) + list(sim.parameters()), g32, g64):
        if a.grad is None or b is None:
            continue
        ga = a.grad.detach().cpu().double()
        e32.append((n, float((ga - b).abs().max() / (c.abs().max() + 1e-30)), float((b - c).abs().max() / (c.abs().max() + 1e-30))))
    # the Gaussian whose scaling gradient is furthest from fp64
    gs, gc = pc._scaling.grad.detach().cpu().double(), pc_c._scaling.grad
    j = int((gs - gc).abs().max(1).values.argmax())
    sc_j = pc._scaling[j].detach().exp().cpu().numpy()
    print(f"   hip-vs-oracle32 / oracle32-vs-64: " + " ".join(f"{n}:{a:.1e}/{b:.1e}" for n, a, b in e32[:7]) +
          f" | worst Gaussian {j}: scales {sc_j} aniso {sc_j.max() / sc_j.min():.1f} opacity {float(torch.sigmoid(pc._opacity[j])):.3f} "
          f"grad hip {gs[j].numpy()} fp64 {gc[j].numpy()} o32 {g32[5][j].numpy()}")
    worst = max(errs, key=lambda e: e[1])
    print(f"step {it:4d} psnr {psnr:7.3f} loss hip {float(lg):.6f} cpu {float(lc):.6f} image max|diff| {float((img_g.detach().cpu().double() - img_c.detach()).abs().max()):.2e} "
          f"worst grad {worst[0]} {worst[1]:.2e} | " + " ".join(f"{n}:{e:.1e}" for n, e in errs[:7]), flush=True)
    for p in list(pc.parameters()) + list(sim.parameters()):
        p.grad = None


t.STEP_HOOK[0] = hook
g, _ = t.run_parity(False, steps, hip_only=True)
