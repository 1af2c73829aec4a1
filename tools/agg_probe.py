import sys, time, torch
sys.path.insert(0, "/root/repo/cloth-splatting_amd"); sys.path.insert(0, "/root/repo")
import os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT + "/cloth-splatting_amd")
import torch.nn as nn
from meshnet.graph_ops import GraphCSR, linear128_agg, linear128
from types import SimpleNamespace
dev = torch.device("cuda")
N, deg = 10000, 30
dst = torch.arange(N, device=dev).repeat_interleave(deg)
src = torch.randint(0, N, (dst.numel(),), device=dev)
order = torch.argsort(src * N + dst)
ei = torch.stack([src[order], dst[order]])
E = ei.shape[1]
csr = GraphCSR.get(ei, N); plan = csr.agg_plan()
A = torch.randn(E, 128, device=dev); lin = nn.Linear(128, 128).to(dev); ln = nn.LayerNorm(128).to(dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    out = torch.empty_like(A)
    print("plain layer      %.1f us" % t(lambda: linear128(A, lin.weight, lin.bias, relu=True, out=out)))
    print("LN layer         %.1f us" % t(lambda: linear128(A, lin.weight, lin.bias, layer_norm=ln, out=out)))
    print("agg layer        %.1f us (incl. zero fill)" % t(lambda: linear128_agg(A, lin.weight, lin.bias, ln, plan, N)))
    p2 = SimpleNamespace(perm=plan.perm, ei=plan.ei, slot_of_row=plan.slot_of_row, slot_dst=torch.full_like(plan.slot_dst, -1))
    print("agg, no atomics  %.1f us" % t(lambda: linear128_agg(A, lin.weight, lin.bias, ln, p2, N)))
    p3 = SimpleNamespace(perm=plan.perm, ei=plan.ei, slot_of_row=torch.full_like(plan.slot_of_row, 255), slot_dst=plan.slot_dst)
    print("agg, S = 0       %.1f us" % t(lambda: linear128_agg(A, lin.weight, lin.bias, ln, p3, N)))
    print("zeros            %.1f us" % t(lambda: torch.zeros(N, 128, device=dev)))
