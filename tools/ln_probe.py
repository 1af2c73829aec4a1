import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_knn_gnn_gpu as T
from meshnet import graph_network as GN, graph_ops as GO
from meshnet.graph_network import EncodeProcessDecode
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 333
N, ei_np = T._irregular_graph(seed)
E = ei_np.shape[1]
torch.manual_seed(seed)
net = EncodeProcessDecode(8, 3, 4, 128, 3, 2, 128).cuda()
gen = torch.Generator().manual_seed(seed)
x = torch.randn(N, 8, generator=gen).cuda().requires_grad_()
e = torch.randn(E, 4, generator=gen).cuda().requires_grad_()
ei = torch.tensor(ei_np, device="cuda")
w = torch.randn(N, 3, generator=gen)
rec = []
orig = GN.layer_norm_rows
def spy(xin, ln):
    y = orig(xin, ln)
    item = {"x": xin.detach().clone(), "ln": ln}
    if y.requires_grad:
        y.register_hook(lambda g, item=item: item.__setitem__("g", g.detach().clone()))
    rec.append(item)
    return y
GN.layer_norm_rows = spy
y = net(x, ei, e)
(y * w.cuda()).sum().backward()
for i, it in enumerate(rec):
    if "g" not in it: continue
    xi, g, ln = it["x"], it["g"], it["ln"]
    def run(dtype, fn):
        xx = xi.to(dtype).clone().requires_grad_()
        ga, be = ln.weight.detach().to(dtype).clone().requires_grad_(), ln.bias.detach().to(dtype).clone().requires_grad_()
        yy = fn(xx, ga, be)
        yy.backward(g.to(dtype))
        return yy.detach().double(), xx.grad.double(), ga.grad.double(), be.grad.double()
    ex = run(torch.float64, lambda a, b, c: torch.nn.functional.layer_norm(a, (128,), b, c, ln.eps))
    th = run(torch.float32, lambda a, b, c: torch.nn.functional.layer_norm(a, (128,), b, c, ln.eps))
    ours = run(torch.float32, lambda a, b, c: GO.LayerNorm128.apply(a, b, c, ln.eps))
    def re(a, b): return float((a - b).abs().max() / b.abs().max())
    var = xi.double().var(1, unbiased=False)
    print(i, tuple(xi.shape), "min var %.2e max|x| %.2e" % (float(var.min()), float(xi.abs().max())),
          "| ours y %.1e dx %.1e dga %.1e dbe %.1e" % tuple(re(a, b) for a, b in zip(ours, ex)),
          "| torch y %.1e dx %.1e dga %.1e dbe %.1e" % tuple(re(a, b) for a, b in zip(th, ex)))
    # per-row dx error
    d_ours = (ours[1] - ex[1]).abs().max(1).values; d_th = (th[1] - ex[1]).abs().max(1).values
    k = int(d_ours.argmax())
    print("     worst row", k, "var %.3e mean %.3e |g|max %.2e  ours %.2e torch %.2e" % (float(var[k]), float(xi[k].double().mean()), float(g[k].abs().max()), float(d_ours[k]), float(d_th[k])))
