import sys, time, numpy as np
sys.path.insert(0,'./tests'); sys.path.insert(0,'./cloth-splatting_amd'); sys.path.insert(0,'.')
from csplat import synthetic as syn
from oracle import raster_oracle as ro
sc = syn.scene_1(P=5000, W=208, H=208, n_cams=1, grid=24, n_times=4, seed=77)
sc["log_scales"] = sc["log_scales"] + np.log(2.0)
g = syn.gaussians_at(sc); cam = sc["cameras"][0]
for st in ("preprocess","bin","all"):
    t0=time.perf_counter()
    for _ in range(5):
        o = ro.forward(g["means3D"], g["opacities"], cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], 208, 208, sc["bg"], shs=g["shs"], sh_degree=3, scales=g["scales"], rotations=g["rotations"], dtype=np.float64, stages=st)
    print(st, (time.perf_counter()-t0)/5)
t0=time.perf_counter(); gr = ro.backward(o, np.ones((3,208,208))); print("bwd", time.perf_counter()-t0)
