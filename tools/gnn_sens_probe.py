#!/usr/bin/env python3
"""tools/gnn_sens_probe.py [seeds...] -- which piece of the training path moves the irregular-graph gradients away from fp64?
Runs the fuzz graphs of tests/test_knn_gnn_gpu.py with pieces of the product path swapped for their plain torch spelling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_knn_gnn_gpu as T
from meshnet import graph_network as GN, graph_ops as GO
from meshnet.graph_network import EncodeProcessDecode
from util import rel_err

def composed(n_, x_, ei_, e_):
    h, ee = n_._encoder(x_, e_)
    for g_ in n_._processor.gnn_stacks:
        m = g_.edge_fn(torch.cat([h.index_select(0, ei_[1]), h.index_select(0, ei_[0]), ee], -1))
        agg = torch.zeros_like(h).index_add_(0, ei_[1], m)
        h = g_.node_fn(torch.cat([agg, h], -1)) + h
        ee = ee + ee
    return n_._decoder(h)

orig = dict(fast=GO.SplitKLinear._fast, ln=GN.layer_norm_rows, lr=GN.linear_rows, l128=GO.linear128)
def plain_l128(A, weight, bias=None, alpha=1.0, relu=False, gather=None, layer_norm=None, add_pre=None, add_post=None, out=None):
    assert gather is None and layer_norm is None and add_pre is None and add_post is None and out is None
    y = (A @ weight.t()) * alpha
    if bias is not None: y = y + bias
    return y.relu() if relu else y
variants = {
    "default": {},
    "torch LN": dict(ln=lambda x, ln: ln(x)),
    "torch linears": dict(lr=lambda x, w, b, min_rows=0, relu=False: (torch.nn.functional.linear(x, w, b).relu() if relu else torch.nn.functional.linear(x, w, b))),
    "fp32 GEMM in SplitK": dict(l128=plain_l128),
}
for seed in [int(a) for a in sys.argv[1:]] or [310, 323, 333]:
    N, ei_np = T._irregular_graph(seed)
    E = ei_np.shape[1]
    torch.manual_seed(seed)
    net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).cuda()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 8, generator=gen).cuda().requires_grad_()
    e = torch.randn(E, 4, generator=gen).cuda().requires_grad_()
    ei = torch.tensor(ei_np, device="cuda")
    w = torch.randn(N, 3, generator=gen)
    n64 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).double()
    n64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    x64, e64 = x.detach().cpu().double().requires_grad_(), e.detach().cpu().double().requires_grad_()
    y64 = composed(n64, x64, torch.tensor(ei_np), e64)
    (y64 * w.double()).sum().backward()
    exact = [x64.grad, e64.grad] + [p.grad for p in n64.parameters()]
    print("seed", seed, "N", N, "E", E, "deg_max", int(np.bincount(ei_np[1]).max()))
    for name, patch in variants.items():
        GO.SplitKLinear._fast = staticmethod(patch.get("fast", orig["fast"]))
        GN.layer_norm_rows = patch.get("ln", orig["ln"]); GN.linear_rows = patch.get("lr", orig["lr"]); GO.linear128 = patch.get("l128", orig["l128"])
        net.zero_grad(); x.grad = None; e.grad = None
        y = net(x, ei, e)
        (y * w.cuda()).sum().backward()
        got = [x.grad, e.grad] + [p.grad for p in net.parameters()]
        errs = [rel_err(a.cpu().numpy(), c.numpy()) for a, c in zip(got, exact) if c is not None and float(c.abs().max()) > 0]
        print("   %-22s fwd %.2e  x %.2e  e %.2e  max-param %.2e" % (name, rel_err(y.detach().cpu().numpy(), y64.detach().numpy()), errs[0], errs[1], max(errs[2:])))
    GO.SplitKLinear._fast = staticmethod(orig["fast"]); GN.layer_norm_rows = orig["ln"]; GN.linear_rows = orig["lr"]; GO.linear128 = orig["l128"]
