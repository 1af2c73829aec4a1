#!/usr/bin/env python3
"""bench_ssim.py: HIP-event timing of the fused image loss (csplat_l1 + csplat_ssim_fwd, csplat_ssim_bwd) on 3x3x800x800."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from csplat import train as tr
x = torch.rand(3, 3, 800, 800, device="cuda", requires_grad=True)
y = torch.rand(3, 3, 800, 800, device="cuda")
opt = SimpleNamespace(lambda_dssim=0.2)
for name, fn in (("fused image loss fwd", lambda: tr.image_losses(x, y, opt)),):
    for _ in range(5):
        l = fn(); l.backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(50):
        x.grad = None
        e[0].record(); l = fn(); e[1].record(); l.backward(); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"{name}: fwd {tf / 50 * 1e3:.1f} us  bwd {tb / 50 * 1e3:.1f} us")
