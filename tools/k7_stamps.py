"""tools/k7_stamps.py -- where a wave of the compositing backward (row form, the batched launch of bench.py) spends its life: s_memtime stamps
left by wave 0 of every (segment, quadrant) workgroup at the phase boundaries (csplat_debug_stamps), reduced to averages over the LIVE
workgroups.  GPU box:  python3 tools/k7_stamps.py"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
from csplat import native, synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_views

dev = torch.device("cuda:0")
P, W, H, V = 100_000, 800, 800, 4
sc = syn.scene_1(P=P, W=W, H=H, n_cams=V)
g = syn.gaussians_at(sc)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731
params = {k: T(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=T(sc["bg"]), scale_modifier=1.0,
                                          viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                                          campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in sc["cameras"]]
buf = torch.zeros(V * 40000 * 12, dtype=torch.int64, device=dev)
dp = torch.randn(V, 3, H, W, device=dev)


def step(stamps):
    for p in params.values():
        p.grad = None
    m2d = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in range(V)]
    colors, _ = rasterize_views(settings, [dict(means3D=params["means3D"], means2D=m2d[i], opacities=params["opacities"], shs=params["shs"],
                                                scales=params["scales"], rotations=params["rotations"]) for i in range(V)], stacked=True)
    native.lib.csplat_debug_stamps(buf.data_ptr() if stamps else None, buf.numel() * 8 if stamps else 0)
    (colors * dp).sum().backward()
    torch.cuda.synchronize()
    native.lib.csplat_debug_stamps(None, 0)


for _ in range(3):
    step(False)
buf.zero_()
step(True)
s = buf.cpu().numpy().reshape(-1, 12).astype(np.int64)
live = s[:, 9] == 1                    # workgroups that reached the end (not the early exits)
started = s[:, 0] != 0
print(f"workgroups: launched with a stamp {int(started.sum())}, live to the end {int(live.sum())}")
L = s[live & (s[:, 8] > 0) & (s[:, 4] != 0)]      # wave 0 had survivors: every stamp was written
names = ["scalar chain (slot -> tile -> range, blk_hi, bbits)", "pixel loads issued, survivor list -> LDS ring", "pixel constants + checkpoint arrive",
         "first records arrive", "the wave's groups"]
d = np.diff(L[:, :6], axis=1)
tot = L[:, 5] - L[:, 0]
print(f"wave 0 of a live workgroup: {tot.mean():.0f} cycles from entry to its last group (median {np.median(tot):.0f}); survivors (padded): {L[:, 8].mean():.0f}")
for k, nme in enumerate(names):
    print(f"  {nme:60s} mean {d[:, k].mean():8.0f}  median {np.median(d[:, k]):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}   ({100 * d[:, k].mean() / tot.mean():.0f} %)")
act = L[L[:, 8] > 0]
da = np.diff(act[:, :6], axis=1)
print(f"waves WITH survivors ({len(act)}): groups phase mean {da[:, 4].mean():.0f} cycles for {act[:, 8].mean() / 4:.1f} groups of four = "
      f"{da[:, 4].sum() / (act[:, 8].sum() / 4):.0f} cycles per group; first-records wait {da[:, 3].mean():.0f}")
# (s_memtime counts per XCD: stamps of DIFFERENT workgroups are not comparable -- only the differences inside one wave are used)
