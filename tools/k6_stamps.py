"""tools/k6_stamps.py -- how long the waves of the compositing forward (one wave per (tile, 4x4 block) item, batched launch of four views) live:
s_memtime at entry / exit, groups of four survivors composited, and the tile's list length, per wave (scratch build with the stamp patch:
csplat_debug_flags bit 21 + csplat_debug_stamps).  GPU box."""
import os, sys
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
T = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)  # noqa: E731
params = {k: T(g[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=T(sc["bg"]), scale_modifier=1.0,
                                          viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                                          campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in sc["cameras"]]
tiles = 50 * 50
total = ((tiles + 7) // 8) * 128
buf = torch.zeros(V * total * 4, dtype=torch.int64, device=dev)


def fwd():
    with torch.no_grad():
        m2d = [torch.zeros(P, 3, device=dev) for _ in range(V)]
        rasterize_views(settings, [dict(means3D=params["means3D"], means2D=m2d[i], opacities=params["opacities"], shs=params["shs"],
                                        scales=params["scales"], rotations=params["rotations"]) for i in range(V)], stacked=True)
    torch.cuda.synchronize()


for _ in range(3):
    fwd()
FL = int(os.environ.get('K6_FLAGS', '0'))
native.lib.csplat_debug_flags((1 << 22) | FL)
native.lib.csplat_debug_stamps(buf.data_ptr(), buf.numel() * 8)
fwd()
native.lib.csplat_debug_stamps(None, 0)
native.lib.csplat_debug_flags(0)
s = buf.cpu().numpy().reshape(-1, 4).astype(np.int64)
s = s[s[:, 0] != 0]
dur = s[:, 1] - s[:, 0]
busy = s[s[:, 2] > 0]
db = busy[:, 1] - busy[:, 0]
print(f"waves with a stamp {len(s)}, with work {len(busy)}; launch span {(s[:, 1].max() - s[:, 0].min()) / 100:.1f} us (s_memtime at 100 MHz)")
print(f"cycles (100 MHz ticks) per busy wave: mean {db.mean():.0f} median {np.median(db):.0f} p90 {np.percentile(db, 90):.0f} p99 {np.percentile(db, 99):.0f} max {db.max()}")
print(f"groups per busy wave: mean {busy[:, 2].mean():.0f} median {np.median(busy[:, 2]):.0f} p99 {np.percentile(busy[:, 2], 99):.0f} max {busy[:, 2].max()};  ticks per group (p50) {np.median(db / busy[:, 2]):.2f}")
order = np.argsort(-db)[:8]
print("longest waves: (ticks, groups, list length, start offset in ticks)", [(int(db[i]), int(busy[i, 2]), int(busy[i, 3]), int(busy[i, 0] - s[:, 0].min())) for i in order])
raw = buf.cpu().numpy().reshape(V, total, 4).astype(np.int64)
for x in range(2):                     # one XCD's waves of view 0: s_memtime is comparable inside an XCD
    r = raw[0][x::8]
    r = r[r[:, 0] != 0]
    r = r[r[:, 0] > np.median(r[:, 0]) - 2_000_000]          # (the stamped launch only)
    t0 = r[:, 0].min()
    e = r[r[:, 2] == 0]
    b = r[r[:, 2] > 0]
    print(f"XCD {x}: waves {len(r)} (busy {len(b)}), first start..last end {(r[:, 1].max() - t0)} ticks; empty waves: mean life {(e[:, 1] - e[:, 0]).mean():.0f}, "
          f"start times p10/p50/p90/max {np.percentile(e[:, 0] - t0, [10, 50, 90, 100]).astype(int).tolist()}; busy start p10/p50/p90/max {np.percentile(b[:, 0] - t0, [10, 50, 90, 100]).astype(int).tolist()}, busy end p50/max {np.percentile(b[:, 1] - t0, [50, 100]).astype(int).tolist()}")
print(f"sum of busy-wave ticks / (8192 slots): {db.sum() / 8192:.0f} ticks = what perfect packing at full occupancy would take")
