"""tests/golden/make_golden.py -- generates the golden vectors under tests/golden/*.npz.

Run ONLY in the build container (it imports the reference from /root/reference, which does not exist
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it pins; every fixture holds inputs + the reference's outputs, no source text:
  camera.npz      scene_reconstruction.cameras.Camera matrices, utils.graphics_utils.getProjectionMatrix /
                  getWorld2View2, for seeded poses incl. a Blender transform_matrix pushed through the
                  c2w[:3,1:3]*=-1 path (dataset_readers.py:352-359)
  sh.npz          utils.sh_utils.eval_sh deg 0..3 + the clamp_min(+0.5) of gaussian_renderer/__init__.py:143
  misc.npz        utils.image_utils.psnr, utils.general_utils.get_expon_lr_func, build_rotation
  gnn.npz         meshnet.graph_network.EncodeProcessDecode / InteractionNetwork and
                  meshnet.cloth_network.ClothMeshSimulator run under a MessagePassing SHIM ("shim-derived":
                  torch_geometric is not installed; the shim implements the documented PyG semantics,
                  SURVEY.md A.3: x_j = x[ei[0]], x_i = x[ei[1]], sum-aggregate over ei[1], update() gets the
                  ORIGINAL propagate kwargs)
  simulator.npz   SinusoidalEncoder / ResidualMeshSimulator / ResidualMeshSimulatorEmbedding.  The file
                  meshnet/meshnet_network.py does not parse (merge-conflict markers); the classes
                  are exec'd at generation time from the reference text with the conflict resolved to the
                  `9b63d7a` side.  Only tensors are stored.
  normalizer.npz  meshnet.model_utils.Normalizer accumulate / normalise / inverse
"""
import inspect
import os
import re
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def install_pyg_shim():
    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr="add", flow="source_to_target"):
            super().__init__()
            self.aggr = aggr

        def propagate(self, edge_index, size=None, **kwargs):
            msg_params = list(inspect.signature(self.message).parameters)
            upd_params = list(inspect.signature(self.update).parameters)[1:]
            n = kwargs["x"].shape[0]
            margs = {}
            for p in msg_params:
                if p.endswith("_i"):
                    margs[p] = kwargs[p[:-2]].index_select(0, edge_index[1])
                elif p.endswith("_j"):
                    margs[p] = kwargs[p[:-2]].index_select(0, edge_index[0])
                else:
                    margs[p] = kwargs[p]
            out = self.message(**margs)
            assert self.aggr == "add"
            agg = torch.zeros(n, out.shape[1], dtype=out.dtype).index_add_(0, edge_index[1], out)
            return self.update(agg, **{p: kwargs[p] for p in upd_params})

    tg = types.ModuleType("torch_geometric")
    tg.nn = types.ModuleType("torch_geometric.nn")
    tg.nn.conv = types.ModuleType("torch_geometric.nn.conv")
    tg.nn.conv.MessagePassing = MessagePassing
    tg.data = types.ModuleType("torch_geometric.data")

    class Data:  # only constructed by helper functions we do not call
        def __init__(self, **kw):
            self.__dict__.update(kw)

    tg.data.Data = Data
    for name, mod in [("torch_geometric", tg), ("torch_geometric.nn", tg.nn), ("torch_geometric.nn.conv", tg.nn.conv),
                      ("torch_geometric.data", tg.data)]:
        sys.modules[name] = mod


def npy(t):
    return t.detach().cpu().numpy()


def gen_camera():
    from scene_reconstruction.cameras import Camera
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    rng = np.random.default_rng(1)
    out = {}
    n = 6
    Rs, Ts, fx, fy, wv, fp, cc, pm, c2ws = [], [], [], [], [], [], [], [], []
    for k in range(n):
        # random camera-to-world (Blender convention): rotation from QR, translation radius ~4
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        c2w = np.eye(4)
        c2w[:3, :3] = Q
        c2w[:3, 3] = rng.normal(size=3) * 2.0 + np.array([0, 0, 3.0])
        c2ws.append(c2w.copy())
        m = c2w.copy()
        m[:3, 1:3] *= -1
        w2c = np.linalg.inv(m)
        R = np.transpose(w2c[:3, :3])
        T = w2c[:3, 3]
        fovx = 0.4 + 0.1 * k
        fovy = 0.35 + 0.12 * k
        img = torch.zeros(3, 8 + k, 10 + k)
        cam = Camera(colmap_id=k, R=R, T=T, FoVx=fovx, FoVy=fovy, image=img, gt_alpha_mask=None, image_name="x",
                     uid=k, data_device="cpu", time=0.1 * k)
        Rs.append(R); Ts.append(T); fx.append(fovx); fy.append(fovy)
        wv.append(npy(cam.world_view_transform)); fp.append(npy(cam.full_proj_transform))
        cc.append(npy(cam.camera_center)); pm.append(npy(cam.projection_matrix))
    out.update(c2w=np.stack(c2ws), R=np.stack(Rs), T=np.stack(Ts), fovx=np.array(fx), fovy=np.array(fy),
               world_view_transform=np.stack(wv), full_proj_transform=np.stack(fp), camera_center=np.stack(cc),
               projection_matrix_T=np.stack(pm))
    out["proj_0p01_100"] = npy(getProjectionMatrix(0.01, 100.0, 0.6911, 0.5))
    out["w2v2_translate"] = getWorld2View2(Rs[0], Ts[0], np.array([0.1, -0.2, 0.3]), 1.5)
    np.savez(os.path.join(OUT, "camera.npz"), **out)


def gen_sh():
    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    g = torch.Generator().manual_seed(2)
    sh = torch.randn(64, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=1)
    out = dict(sh=npy(sh), dirs=npy(dirs))
    for deg in range(4):
        r = eval_sh(deg, sh, dirs)
        out[f"rgb_deg{deg}"] = npy(r)
        out[f"clamped_deg{deg}"] = npy(torch.clamp_min(r + 0.5, 0.0))
    x = torch.rand(10, 3, generator=g)
    out["rgb2sh_in"] = npy(x); out["rgb2sh"] = npy(RGB2SH(x)); out["sh2rgb"] = npy(SH2RGB(x))
    np.savez(os.path.join(OUT, "sh.npz"), **out)


def gen_misc():
    from utils.image_utils import psnr
    from utils.general_utils import get_expon_lr_func, inverse_sigmoid
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 16, 16, generator=g)
    out = dict(psnr_a=npy(a), psnr_b=npy(b), psnr=npy(psnr(a, b)))
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=20000)
    steps = np.array([0, 1, 10, 100, 1000, 5000, 19999, 20000, 30000])
    out["lr_steps"] = steps; out["lr"] = np.array([f(int(s)) for s in steps])
    f2 = get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=3000)
    out["lr2"] = np.array([f2(int(s)) for s in steps])
    x = torch.rand(16, generator=g) * 0.98 + 0.01
    out["inv_sigmoid_in"] = npy(x); out["inv_sigmoid"] = npy(inverse_sigmoid(x))
    # build_rotation allocates on 'cuda' in the reference; restate through its formula on CPU is not "reference
    # output", so it is NOT stored here (oracle/raster_ref.c quat_to_rot cites the lines instead).
    np.savez(os.path.join(OUT, "misc.npz"), **out)


def gen_gnn():
    install_pyg_shim()
    from meshnet.graph_network import EncodeProcessDecode, InteractionNetwork
    from meshnet.cloth_network import ClothMeshSimulator
    torch.manual_seed(0)
    N, E = 50, 300
    g = torch.Generator().manual_seed(4)
    ei = torch.randint(0, N, (2, E), generator=g)
    # --- single InteractionNetwork (F7: edge output is 2 x edge input)
    inet = InteractionNetwork(nnode_in=16, nnode_out=16, nedge_in=16, nedge_out=16, nmlp_layers=2, mlp_hidden_dim=16)
    x0, e0 = torch.randn(N, 16, generator=g), torch.randn(E, 16, generator=g)
    x1, e1 = inet(x0, ei, e0)
    out = dict(edge_index=npy(ei), in_x=npy(x0), in_e=npy(e0), in_x_out=npy(x1), in_e_out=npy(e1))
    for k, v in inet.state_dict().items():
        out["inet." + k] = npy(v)
    # --- EncodeProcessDecode
    net = EncodeProcessDecode(nnode_in_features=8, nnode_out_features=3, nedge_in_features=4, latent_dim=32,
                              nmessage_passing_steps=3, nmlp_layers=2, mlp_hidden_dim=32)
    x = torch.randn(N, 8, generator=g)
    ef = torch.randn(E, 4, generator=g)
    y = net(x, ei, ef)
    out.update(epd_x=npy(x), epd_e=npy(ef), epd_y=npy(y))
    for k, v in net.state_dict().items():
        out["epd." + k] = npy(v)
    # gradient of sum(y * w) wrt inputs and one weight (pins the autograd twins)
    xg, eg = x.clone().requires_grad_(True), ef.clone().requires_grad_(True)
    wgt = torch.randn(N, 3, generator=g)
    (net(xg, ei, eg) * wgt).sum().backward()
    out.update(epd_w=npy(wgt), epd_dx=npy(xg.grad), epd_de=npy(eg.grad),
               epd_dW_first=npy(net._processor.gnn_stacks[0].edge_fn[0][0].weight.grad))
    # --- ClothMeshSimulator (normalize=True), train-mode accumulate then eval rollout step
    torch.manual_seed(1)
    sim = ClothMeshSimulator(simulation_dimensions=3, nnode_in=8, nedge_in=4, latent_dim=32, nmessage_passing_steps=2,
                             nmlp_layers=2, mlp_hidden_dim=32, nnode_types=2, node_type_embedding_size=2,
                             normalize=True, device="cpu")
    vel = torch.randn(N, 6, generator=g) * 0.1
    ntype = torch.randint(0, 2, (N, 1), generator=g)
    tgt = torch.randn(N, 3, generator=g) * 0.1
    noise = torch.randn(N, 6, generator=g) * 0.01
    sim.train()
    pa, ta = sim.predict_acceleration(vel, ntype, ei, ef, target_velocities=tgt, velocity_noise=noise)
    sim.eval()
    pv = sim.predict_velocity(vel, ntype, ei, ef)
    out.update(sim_vel=npy(vel), sim_type=npy(ntype), sim_tgt=npy(tgt), sim_noise=npy(noise), sim_pred_acc=npy(pa),
               sim_tgt_acc=npy(ta), sim_pred_vel=npy(pv))
    for k, v in sim.state_dict().items():
        out["sim." + k] = npy(v)
    for nm in ("_output_normalizer", "_node_normalizer"):
        for k, v in getattr(sim, nm).get_variable().items():
            if torch.is_tensor(v):
                out[f"sim{nm}.{k}"] = npy(v)
    # identity-normaliser flavour (normalize=False is the train_meshnet_sim.py default path)
    torch.manual_seed(2)
    sim2 = ClothMeshSimulator(3, 8, 4, 32, 2, 2, 32, 2, 2, normalize=False, device="cpu")
    sim2.eval()
    out["sim2_pred_vel"] = npy(sim2.predict_velocity(vel, ntype, ei, ef))
    for k, v in sim2.state_dict().items():
        out["sim2." + k] = npy(v)
    np.savez(os.path.join(OUT, "gnn.npz"), **out)


def gen_normalizer():
    install_pyg_shim()
    from meshnet.model_utils import Normalizer
    g = torch.Generator().manual_seed(5)
    nz = Normalizer(size=5, device="cpu")
    b1, b2 = torch.randn(20, 5, generator=g) * 3 + 1, torch.randn(30, 5, generator=g) * 0.5 - 2
    o1 = nz(b1, True)
    o2 = nz(b2, True)
    o3 = nz(b1, False)
    inv = nz.inverse(o3)
    np.savez(os.path.join(OUT, "normalizer.npz"), b1=npy(b1), b2=npy(b2), o1=npy(o1), o2=npy(o2), o3=npy(o3),
             inv=npy(inv), acc_sum=npy(nz._acc_sum), acc_sum_squared=npy(nz._acc_sum_squared),
             acc_count=npy(nz._acc_count), num_acc=npy(nz._num_accumulations))


def gen_simulator():
    """exec the parseable class bodies of meshnet/meshnet_network.py with the conflict resolved to 9b63d7a."""
    src = open(os.path.join(REF, "meshnet/meshnet_network.py")).read()
    # resolve conflict blocks: keep the part between '=======' and '>>>>>>>'
    src = re.sub(r"<<<<<<< HEAD\n.*?=======\n(.*?)>>>>>>> [^\n]*\n", r"\1", src, flags=re.S)
    start = src.index("class SinusoidalEncoder")
    ns = {"torch": torch, "nn": torch.nn, "Optional": __import__("typing").Optional, "np": np}
    exec(compile(src[start:], "<meshnet_network tail>", "exec"), ns)
    Sin, Res, Emb = ns["SinusoidalEncoder"], ns["ResidualMeshSimulator"], ns["ResidualMeshSimulatorEmbedding"]
    g = torch.Generator().manual_seed(6)
    out = {}
    enc = Sin(input_dim=1, num_freqs=6, device="cpu")
    t = torch.tensor([0.0, 0.1, 0.3333, 0.9, 1.0])[:, None]
    out["enc_in"] = npy(t); out["enc_out"] = npy(torch.stack([enc(ti) for ti in t]))
    out["enc_dim"] = np.array(enc.output_dim)
    enc3 = Sin(input_dim=3, num_freqs=4, min_freq_log2=-1, scale=0.5, use_identity=False)
    x3 = torch.randn(7, 3, generator=g)
    out["enc3_in"] = npy(x3); out["enc3_out"] = npy(enc3(x3))
    Tn, V = 5, 11
    mp = torch.randn(Tn, V, 3, generator=g)
    torch.manual_seed(7)
    sim = Res(mp, device="cpu")
    with torch.no_grad():
        sim.output.weight.normal_(0, 0.05, generator=g)  # make the residual visible in the fixture
        sim.output.bias.normal_(0, 0.05, generator=g)
    out["res_mesh"] = npy(mp)
    for k, v in sim.state_dict().items():
        out["res." + k] = npy(v)
    times = [0.0, 0.25, 0.5, 0.74, 1.0]
    out["res_times"] = np.array(times, np.float32)
    out["res_out"] = np.stack([npy(sim(torch.tensor(tt, dtype=torch.float32).repeat(V, 1))) for tt in times])
    out["res_time_delta"] = np.array(sim.time_delta)
    sim1 = Res(mp[:1], device="cpu")  # n_times == 1 branch of the 9b63d7a side
    out["res1_time_delta"] = np.array(sim1.time_delta)
    try:
        sim(torch.tensor(1.3).repeat(V, 1))
        out["res_oob_raises"] = np.array(0)
    except ValueError:
        out["res_oob_raises"] = np.array(1)
    torch.manual_seed(8)
    emb = Emb(mp, device="cpu")
    for k, v in emb.state_dict().items():
        out["emb." + k] = npy(v)
    out["emb_out"] = np.stack([npy(emb(torch.tensor(tt, dtype=torch.float32).repeat(V, 1))) for tt in times])
    np.savez(os.path.join(OUT, "simulator.npz"), **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    gen_camera(); gen_sh(); gen_misc(); gen_normalizer(); gen_gnn(); gen_simulator()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
