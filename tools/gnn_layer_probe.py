#!/usr/bin/env python3
"""tools/gnn_layer_probe.py [seed] -- per-layer forward / gradient error of the training path against fp64 on a fuzz graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_knn_gnn_gpu as T
from meshnet.graph_network import EncodeProcessDecode
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 323
N, ei_np = T._irregular_graph(seed)
E = ei_np.shape[1]
torch.manual_seed(seed)
net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).cuda()
gen = torch.Generator().manual_seed(seed)
x = torch.randn(N, 8, generator=gen).cuda().requires_grad_()
e = torch.randn(E, 4, generator=gen).cuda().requires_grad_()
ei = torch.tensor(ei_np, device="cuda")
w = torch.randn(N, 3, generator=gen)

def run(n_, x_, ei_, e_, w_, product):
    keep = {}
    def k(name, t):
        if t.requires_grad: t.retain_grad()
        keep[name] = t
        return t
    h, ee = n_._encoder(x_, e_)
    k("enc.h", h); k("enc.e", ee)
    scale = 1.0
    for li, g_ in enumerate(n_._processor.gnn_stacks):
        if product:
            h, ee = g_.message_update(h, ei_, ee, scale); scale *= 2.0
        else:
            pre = torch.cat([h.index_select(0, ei_[1]), h.index_select(0, ei_[0]), ee * scale], -1)
            mlp = g_.edge_fn[0]
            z0 = k("L%d.edge.z0" % li, mlp[0](pre))
            z1 = k("L%d.edge.z1" % li, mlp[2](mlp[1](z0)))
            z2 = k("L%d.edge.z2" % li, mlp[4](mlp[3](z1)))
            m = k("L%d.msg" % li, g_.edge_fn[1](mlp[5](z2)))
            agg = k("L%d.agg" % li, torch.zeros_like(h).index_add_(0, ei_[1], m))
            mn = g_.node_fn[0]
            y0 = k("L%d.node.z0" % li, mn[0](torch.cat([agg, h], -1)))
            y1 = k("L%d.node.z1" % li, mn[2](mn[1](y0)))
            y2 = k("L%d.node.z2" % li, mn[4](mn[3](y1)))
            h = g_.node_fn[1](mn[5](y2)) + h
            scale *= 2.0
        k("L%d.h" % li, h)
    y = k("y", n_._decoder(h))
    (y * w_).sum().backward()
    return keep

ours = run(net, x, ei, e, w.cuda(), True)
n64 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).double()
n64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
x64, e64 = x.detach().cpu().double().requires_grad_(), e.detach().cpu().double().requires_grad_()
ex = run(n64, x64, torch.tensor(ei_np), e64, w.double(), False)
n32 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128)
n32.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
x32, e32 = x.detach().cpu().requires_grad_(), e.detach().cpu().requires_grad_()
pl = run(n32, x32, torch.tensor(ei_np), e32, w, False)
def re(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
print("seed", seed, "N", N, "E", E)
for name in ours:
    print("%-8s fwd ours %.1e plain %.1e | grad ours %.1e plain %.1e" % (name, re(ours[name].detach(), ex[name].detach()), re(pl[name].detach(), ex[name].detach()),
          re(ours[name].grad, ex[name].grad), re(pl[name].grad, ex[name].grad)))
# near-zero pre-activations in the fp64 run, weighted by how much gradient passes through the unit
print("closest ReLU pre-activations in fp64 (|z| / row max |z|), with |dL/dz| of that unit relative to the tensor's max:")
for name, t in ex.items():
    if ".z" in name and not name.endswith("z2"):
        z = t.detach(); g = t.grad
        rel = z.abs() / z.abs().max(1, keepdim=True).values
        imp = g.abs() / g.abs().max()
        # a unit that is off has zero gradient: use the upstream side instead -> rank by closeness only, report top 3
        idx = torch.topk(rel.flatten(), 3, largest=False).indices
        print("  ", name, [(int(i // 128), "%.1e" % float(rel.flatten()[i]), "%.1e" % float(z.flatten()[i])) for i in idx])

# ---- one layer under the microscope: the layer whose backward moves the gradients
from meshnet import graph_network as GN
from meshnet.graph_ops import EdgeCombine, GraphCSR, SegmentSum, layer_norm_rows, linear_rows
import torch.nn as nn
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g_ = net._processor.gnn_stacks[L]
hin = (ex["L%d.h" % (L - 1)] if L else ex["enc.h"]).detach().float().cuda().requires_grad_()
eb = ex["enc.e"].detach().float().cuda().requires_grad_()
scale = 2.0 ** L
csr = GraphCSR.get(ei, N)
n = 128
mlp_e = g_.edge_fn[0]; W = mlp_e[0].weight
xa = hin @ W[:, :n].t(); xb = hin @ W[:, n:2 * n].t()
ec = linear_rows(eb, W[:, 2 * n:] * scale, mlp_e[0].bias)
t = {}
def k(name, v):
    v.retain_grad(); t[name] = v; return v
a0 = k("edge.a0", EdgeCombine.apply(xa, xb, ec, csr, True))          # relu(z0)
z1 = k("edge.z1", linear_rows(a0, mlp_e[2].weight, mlp_e[2].bias))
z2 = k("edge.z2", linear_rows(z1.relu(), mlp_e[4].weight, mlp_e[4].bias))
msg = k("msg", layer_norm_rows(z2, g_.edge_fn[1]))
agg = k("agg", SegmentSum.apply(msg, csr))
mn = g_.node_fn[0]
y0 = k("node.z0", torch.addmm(mn[0].bias, agg, mn[0].weight[:, :n].t()) + hin @ mn[0].weight[:, n:].t())
y1 = k("node.z1", mn[2](y0.relu()))
y2 = k("node.z2", mn[4](y1.relu()))
out = layer_norm_rows(y2, g_.node_fn[1]) + hin
out.backward(ex["L%d.h" % L].grad.float().cuda())
print("layer", L, "step by step (inputs and upstream gradient taken from the fp64 run):")
for name, v in t.items():
    ref = ex["L%d.%s" % (L, name)] if name != "edge.a0" else ex["L%d.edge.z0" % L]
    gref = ref.grad
    if name == "edge.a0":
        print("  %-8s fwd %.1e" % (name, re(v.detach(), ref.detach().relu())), "grad(masked) %.1e" % re(v.grad * (v.detach() > 0), gref))
    else:
        print("  %-8s fwd %.1e grad %.1e" % (name, re(v.detach(), ref.detach()), re(v.grad, gref)))
print("  hin grad %.1e   eb grad %.1e" % (re(hin.grad, (ex["L%d.h" % (L - 1)] if L else ex["enc.h"]).grad - ex["L%d.h" % L].grad * 0), 0.0))
