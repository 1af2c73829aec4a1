"""tools/overlap_probe.py -- does K7 (bound by the memory-side float-atomic rate) leave room on the chip for another view group's forward?
Two independent groups of V/2 views: the backward of group A (K7 + K8) on one stream, the forward of group B (K1..K6) on another, launched
together, against the same two pieces back to back on one stream.  GPU box; prints wall times of the three arrangements."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
from csplat import synthetic as syn  # noqa: E402
from csplat.train import l1_loss  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_views  # noqa: E402

dev = torch.device("cuda:0")
P, W, H, V = 100_000, 800, 800, 4
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731
scene = syn.scene_1(P=P, W=W, H=H, n_cams=V)
g = syn.gaussians_at(scene)
names = ("means3D", "opacities", "shs", "scales", "rotations")
params = {k: T(g[k], True) for k in names}
bg = T(scene["bg"])
cams = [syn.make_camera(-180.0 + 360.0 * k / V, W, H) for k in range(V)]
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
                                          viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                                          campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in cams]
targets = torch.rand(V, 3, H, W, device=dev)
zeros = torch.zeros(V, P, 3, device=dev)
one = torch.ones((), device=dev)


def fwd(idx):
    m2 = [zeros[i].detach().requires_grad_() for i in idx]
    colors, _ = rasterize_views([settings[i] for i in idx], [dict(means3D=params["means3D"], means2D=m2[j], opacities=params["opacities"],
                                                                   shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
                                                              for j, i in enumerate(idx)], stacked=True)
    return l1_loss(colors, targets[idx[0]:idx[-1] + 1])


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


A, B = [0, 1], [2, 3]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    for p_ in params.values():
        p_.grad = None
    la = fwd(A)
    la.backward(gradient=one)
    lb = fwd(B)
    lb.backward(gradient=one)


def serial_all():
    for p_ in params.values():
        p_.grad = None
    l_ = fwd(A + B)
    l_.backward(gradient=one)


def overlapped():
    """fwd A | then bwd A on stream sA together with fwd B on stream sB | then bwd B"""
    for p_ in params.values():
        p_.grad = None
    main = torch.cuda.current_stream()
    la = fwd(A)
    sA.wait_stream(main); sB.wait_stream(main)
    with torch.cuda.stream(sB):
        lb = fwd(B)
    with torch.cuda.stream(sA):
        la.backward(gradient=one)
    main.wait_stream(sA); main.wait_stream(sB)
    lb.backward(gradient=one)


from csplat.graphs import ReplayedSteps  # noqa: E402
for name, fn in (("all four views, one batch", serial_all), ("two groups of two, back to back", serial), ("two groups, bwd(A) || fwd(B)", overlapped)):
    try:
        rs = ReplayedSteps(fn, dev, G=2)
        rs.record()
        ms = timeit(rs.step, 200)
        rs.check()
        print("REPLAYED %-34s %.4f ms" % (name + ":", ms))
    except Exception as e:
        print("REPLAYED", name, "failed:", repr(e)[:300])
print("all four views, one batch (fwd + bwd):        %.4f ms" % timeit(serial_all))
print("two groups of two, back to back:               %.4f ms" % timeit(serial))
print("two groups, bwd(A) || fwd(B) on two streams:   %.4f ms" % timeit(overlapped))
