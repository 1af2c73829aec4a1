import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
from csplat import native, synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_views
flags = int(sys.argv[1]); out = sys.argv[2]
native.lib.csplat_debug_flags(flags)
dev = torch.device("cuda:0")
P, W, H, V = 5000, 208, 208, 3
sc = syn.scene_1(P=P, W=W, H=H, n_cams=V)
g = syn.gaussians_at(sc)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731
params = {k: T(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
with torch.no_grad():
    params["scales"] += 0.7
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=T(sc["bg"]), scale_modifier=1.0,
                                          viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                                          campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in sc["cameras"]]
torch.manual_seed(0)
dp = torch.randn(V, 3, H, W, device=dev)
m2d = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in range(V)]
colors, _ = rasterize_views(settings, [dict(means3D=params["means3D"], means2D=m2d[i], opacities=params["opacities"], shs=params["shs"],
                                            scales=params["scales"], rotations=params["rotations"]) for i in range(V)], stacked=True)
(colors * dp).sum().backward()
torch.cuda.synchronize()
res = {k: v.grad.cpu().numpy() for k, v in params.items()}
res.update({f"m2d{i}": m.grad.cpu().numpy() for i, m in enumerate(m2d)})
res["img"] = colors.detach().cpu().numpy()
np.savez(out, **res)
print("saved")
