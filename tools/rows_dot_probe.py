"""time ShapeSimulator's rows-dot forward (k_rows_dot_fwd) alone: python tools/rows_dot_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cloth-splatting_amd"))
from meshnet import graph_ops as go
dev = torch.device("cuda")
R, T = 30000, 3
W = torch.randn(R, 256, device=dev); b = torch.randn(R, device=dev); h = torch.randn(T, 256, device=dev); base = torch.randn(T, R, device=dev)
for _ in range(5): go._rows_dot_fwd(h, W, b, base)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): go._rows_dot_fwd(h, W, b, base)
torch.cuda.synchronize(); print("rows_dot_fwd %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
