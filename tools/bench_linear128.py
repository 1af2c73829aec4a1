#!/usr/bin/env python3
"""Micro-benchmark of csplat_linear128 at the BASELINE configs[3] edge shape (M = 300,000 rows): per-variant time,
achieved fp32 FLOP/s and HBM GB/s (algorithmic: one read + one write of the activations)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
import torch  # noqa: E402
from meshnet.graph_ops import linear128  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
A = torch.randn(M, 128, device=dev)
W = torch.randn(128, 128, device=dev) * 0.1
b = torch.randn(128, device=dev)
Nn = 10_000
ga, gb = torch.randn(Nn, 128, device=dev), torch.randn(Nn, 128, device=dev)
ia = torch.arange(M, device=dev) // 30
ib = (ia + torch.randint(-64, 65, (M,), device=dev)).clamp_(0, Nn - 1)
ln = torch.nn.LayerNorm(128).to(dev)
out = torch.empty_like(A)
from csplat import native as _n  # noqa: E402
for mode in (0, 1):
  _n.check(_n.lib.csplat_linear128_mode(mode), "mode")
  print("mode", mode, "(0 = fp32 MFMA, 1 = 3-way bf16 split)", flush=True)
  for name, kw in (("plain", {}), ("ln", {"layer_norm": ln}), ("gather", {"gather": (ga, ia, gb, ib)})):
      with torch.no_grad():
          for _ in range(3):
              linear128(A, W, b, relu=True, out=out, **kw)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(reps):
              linear128(A, W, b, relu=True, out=out, **kw)
          e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / reps * 1e3
      print(f"{name:7s} {us:8.1f} us  {2 * M * 128 * 128 / us / 1e6:7.1f} TFLOP/s  {2 * M * 512 / us / 1e3:7.1f} GB/s", flush=True)

_n.check(_n.lib.csplat_linear128_mode(1), "mode")
