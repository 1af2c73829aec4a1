"""tools/gnn_fuzz_debug.py <seed> -- per-parameter gradient errors of test_gnn_irregular_graphs_vs_oracle's network for one seed, with the
training-path kernels switched off one at a time (which piece loses accuracy on an ill-conditioned graph?)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cloth-splatting_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_knn_gnn_gpu as T
from meshnet import graph_network as gn, graph_ops as go

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 305


def run(tag):
    from meshnet.graph_network import EncodeProcessDecode
    N, ei_np = T._irregular_graph(seed)
    E = ei_np.shape[1]
    torch.manual_seed(seed)
    net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).cuda()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 8, generator=gen).cuda().requires_grad_()
    e = torch.randn(E, 4, generator=gen).cuda().requires_grad_()
    ei = torch.tensor(ei_np, device="cuda")
    y = net(x, ei, e)
    w = torch.randn(N, 3, generator=gen)
    (y * w.cuda()).sum().backward()
    got = [x.grad.clone(), e.grad.clone()]

    def composed(n_, x_, ei_, e_):
        h, ee = n_._encoder(x_, e_)
        for g_ in n_._processor.gnn_stacks:
            m = g_.edge_fn(torch.cat([h.index_select(0, ei_[1]), h.index_select(0, ei_[0]), ee], -1))
            agg = torch.zeros_like(h).index_add_(0, ei_[1], m)
            h = g_.node_fn(torch.cat([agg, h], -1)) + h
            ee = ee + ee
        return n_._decoder(h)
    net64 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).double()
    net64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    x64, e64 = x.detach().cpu().double().requires_grad_(), e.detach().cpu().double().requires_grad_()
    (composed(net64, x64, torch.tensor(ei_np), e64) * w.double()).sum().backward()
    r = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())  # noqa: E731
    print(f"{tag:28s} N={N} E={E} deg_max={int(np.bincount(ei_np[1]).max())}  err x {r(got[0], x64.grad):.2e}  e {r(got[1], e64.grad):.2e}  "
          f"y {r(y.detach(), composed(net64, x64, torch.tensor(ei_np), e64).detach()):.2e}")


run("all fused")
orig_ln = gn.layer_norm_rows
gn.layer_norm_rows = lambda x, ln: ln(x)
run("library LayerNorm")
gn.layer_norm_rows = orig_ln
orig_rm = go.relu_mask_bias128
def slow(g, out):
    gm = torch.ops.aten.threshold_backward(g, out, 0) if out is not None else g
    return gm, gm.sum(0)
go.relu_mask_bias128 = slow
run("library relu-mask/bias")
go.relu_mask_bias128 = orig_rm


def conditioning():
    """how much do the fp64 gradients move when every weight is perturbed by 1e-7 relative (one fp32 rounding)?"""
    from meshnet.graph_network import EncodeProcessDecode
    N, ei_np = T._irregular_graph(seed)
    E = ei_np.shape[1]
    torch.manual_seed(seed)
    net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 8, generator=gen)
    e = torch.randn(E, 4, generator=gen)
    w = torch.randn(N, 3, generator=gen)
    ei = torch.tensor(ei_np)

    def composed(n_, x_, ei_, e_):
        h, ee = n_._encoder(x_, e_)
        for g_ in n_._processor.gnn_stacks:
            m = g_.edge_fn(torch.cat([h.index_select(0, ei_[1]), h.index_select(0, ei_[0]), ee], -1))
            agg = torch.zeros_like(h).index_add_(0, ei_[1], m)
            h = g_.node_fn(torch.cat([agg, h], -1)) + h
            ee = ee + ee
        return n_._decoder(h)
    outs = []
    for pert in (0.0, 1e-7):
        n64 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).double()
        sd = {k: v.double() for k, v in net.state_dict().items()}
        g2 = torch.Generator().manual_seed(1)
        if pert:
            sd = {k: v * (1 + pert * torch.randn(v.shape, generator=g2, dtype=torch.float64)) for k, v in sd.items()}
        n64.load_state_dict(sd)
        x64, e64 = x.double().requires_grad_(), e.double().requires_grad_()
        (composed(n64, x64, ei, e64) * w.double()).sum().backward()
        outs.append((x64.grad.clone(), e64.grad.clone()))
    r = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    print(f"conditioning (weights x (1 + 1e-7 randn), fp64): d(grad x) {r(outs[1][0], outs[0][0]):.2e}  d(grad e) {r(outs[1][1], outs[0][1]):.2e}")


conditioning()
