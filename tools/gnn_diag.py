#!/usr/bin/env python3
"""gnn_diag.py [seeds]: input-gradient error of the HIP GNN path vs an fp64 restatement, next to the error of a plain fp32
torch composition (index_select / cat / index_add_) of the same network -- conditioning vs defect."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
import numpy as np
import torch
from meshnet.graph_network import EncodeProcessDecode
src = open(os.path.join(ROOT, "tests", "test_knn_gnn_gpu.py")).read()
ns = {}
exec("import numpy as np\n" + src[src.index("def _irregular_graph(seed):"):src.index('@pytest.mark.parametrize("seed", list(range(300')], ns)


def composed(net, x, ei, e):
    h, ee = net._encoder(x, e)
    for g_ in net._processor.gnn_stacks:
        m = g_.edge_fn(torch.cat([h.index_select(0, ei[1]), h.index_select(0, ei[0]), ee], -1))
        agg = torch.zeros_like(h).index_add_(0, ei[1], m)
        h = g_.node_fn(torch.cat([agg, h], -1)) + h
        ee = ee + ee
    return net._decoder(h)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.abs().max() + 1e-30))


for seed in [int(a) for a in sys.argv[1:]] or range(300, 310):
    N, ei_np = ns["_irregular_graph"](seed)
    E = ei_np.shape[1]
    torch.manual_seed(seed)
    net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).cuda()
    gen = torch.Generator().manual_seed(seed)
    x0 = torch.randn(N, 8, generator=gen); e0 = torch.randn(E, 4, generator=gen); w = None
    ei = torch.tensor(ei_np, device="cuda")
    res = {}
    for name in ("hip", "fp32", "fp64"):
        if name == "fp64":
            n2 = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).double()
            n2.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
            x, e, eidx = x0.double().requires_grad_(), e0.double().requires_grad_(), torch.tensor(ei_np)
            y = composed(n2, x, eidx, e)
        else:
            x, e = x0.cuda().requires_grad_(), e0.cuda().requires_grad_()
            y = net(x, ei, e) if name == "hip" else composed(net, x, ei, e)
        if w is None:
            w = torch.randn(N, 3, generator=gen)
        (y * w.to(y)).sum().backward()
        res[name] = (y.detach(), x.grad, e.grad)
    deg = np.bincount(ei_np[1], minlength=N).max() if E else 0
    print(f"seed {seed}: N={N} E={E} max in-degree {deg} | y: hip {rel(res['hip'][0], res['fp64'][0]):.1e} fp32 {rel(res['fp32'][0], res['fp64'][0]):.1e}"
          f" | dx: hip {rel(res['hip'][1], res['fp64'][1]):.1e} fp32 {rel(res['fp32'][1], res['fp64'][1]):.1e}"
          + (f" | de: hip {rel(res['hip'][2], res['fp64'][2]):.1e} fp32 {rel(res['fp32'][2], res['fp64'][2]):.1e}" if E else ""))
