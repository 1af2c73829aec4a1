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
  densify.npz     MultiGaussianMesh densify / prune / opacity reset / Adam-state surgery / cleanup (gen_densify)
  scene_io.npz    dataset_readers.readCamerasFromTransforms / read_timeline on blender_scene/ (gen_scene_io)
  meshsim.npz     meshnet.meshnet_network.MeshSimulator predict_dx (train, noise) / predict_position (eval) (gen_meshsim)
  mesh_transform.npz  MultiGaussianMesh.get_xyz (+ autograd gradients) and get_rotation with roma served by scipy
  losses.npz      utils.loss_utils.l1_loss / ssim, train_utils.image_losses / regularization, masked and unmasked
  gnn128.npz      EncodeProcessDecode at latent 128 on a tall graph, closed-form weights: forward + gradients (gen_gnn128)
  render_wiring.npz  gaussian_renderer.render itself with a RECORDING stand-in for the rasterizer extension (gen_render_wiring)
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


def closed_form_weights(module, salt=0):
    """fill every parameter of `module` with values that any host can regenerate exactly: an integer hash of (parameter index,
    element index) mapped to [-0.5, 0.5) in float64, scaled by 2 / sqrt(fan_in) (LayerNorm: 1 + 0.2 h / 0.2 h), cast to float32.
    Lets a fixture pin a 128-wide network WITHOUT storing its ~0.5 M weights; tests/util.py holds the identical function."""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            i = np.arange(p.numel(), dtype=np.uint64)
            h = ((i * np.uint64(2654435761) + np.uint64(40503 * (k + 1) + 977 * salt)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32) - 0.5
            if p.dim() == 2:
                v = h * (2.0 / np.sqrt(p.shape[1]))
            elif name.endswith("weight"):           # LayerNorm gain
                v = 1.0 + 0.2 * h
            else:
                v = 0.2 * h
            p.copy_(torch.from_numpy(v.astype(np.float32)).reshape(p.shape))
    return module


def gen_gnn128():
    """EncodeProcessDecode at the config-4 width (latent 128, 2 hidden layers of 128) on a TALL graph (E = 16,640 >= the row count
    from which the build's training path runs its fused MFMA autograd nodes; rollout kernels have no row threshold): the
    reference's own module under the PyG shim, weights in closed form (closed_form_weights), forward output, and the gradients of
    sum(y * w) w.r.t. the node / edge inputs, one edge-MLP weight of every kind (first-layer block, hidden, last), a node-MLP weight,
    a LayerNorm gain and a bias."""
    install_pyg_shim()
    from meshnet.graph_network import EncodeProcessDecode
    g = torch.Generator().manual_seed(128)
    N, deg = 520, 32
    E = N * deg
    dst = torch.arange(N).repeat_interleave(deg)
    src = (dst + torch.randint(1, 40, (E,), generator=g)) % N
    perm = torch.randperm(E, generator=g)
    ei = torch.stack([src, dst])[:, perm].contiguous()
    net = closed_form_weights(EncodeProcessDecode(nnode_in_features=8, nnode_out_features=3, nedge_in_features=4, latent_dim=128,
                                                  nmessage_passing_steps=2, nmlp_layers=2, mlp_hidden_dim=128))
    x = torch.randn(N, 8, generator=g)
    ef = torch.randn(E, 4, generator=g)
    w = torch.randn(N, 3, generator=g)
    with torch.no_grad():
        y_eval = net(x, ei, ef)
    xg, eg = x.clone().requires_grad_(True), ef.clone().requires_grad_(True)
    y = net(xg, ei, eg)
    (y * w).sum().backward()
    l0 = net._processor.gnn_stacks[0]
    l1 = net._processor.gnn_stacks[1]
    out = dict(edge_index=npy(ei).astype(np.int32), x=npy(x), e=npy(ef), w=npy(w), y=npy(y), y_eval=npy(y_eval), dx=npy(xg.grad),
               de=npy(eg.grad),
               dW_edge_first0=npy(l0.edge_fn[0][0].weight.grad), dW_edge_hidden0=npy(l0.edge_fn[0][2].weight.grad),
               dW_edge_last1=npy(l1.edge_fn[0][4].weight.grad), db_edge_last1=npy(l1.edge_fn[0][4].bias.grad),
               dW_node_first1=npy(l1.node_fn[0][0].weight.grad), dgamma_edge0=npy(l0.edge_fn[1].weight.grad),
               dbeta_node1=npy(l1.node_fn[1].bias.grad), dW_enc_edge=npy(net._encoder.edge_fn[0][0].weight.grad),
               dW_dec_last=npy(net._decoder.node_fn[4].weight.grad),
               w_probe=npy(l0.edge_fn[0][2].weight[:2, :5]))         # a few weights, so that the test can check its regeneration
    # the same network in float64 (same weights): how far the reference's OWN fp32 gradients are from exact arithmetic -- ~10^6 ReLU
    # pre-activations, some within fp32 rounding of zero, make the gradient a discontinuous function of the rounding; the test holds
    # the build to the larger of 1e-4 and a small multiple of this distance
    net64 = EncodeProcessDecode(nnode_in_features=8, nnode_out_features=3, nedge_in_features=4, latent_dim=128,
                                nmessage_passing_steps=2, nmlp_layers=2, mlp_hidden_dim=128).double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    xg6, eg6 = x.double().requires_grad_(True), ef.double().requires_grad_(True)
    y6 = net64(xg6, ei, eg6)
    (y6 * w.double()).sum().backward()
    m0, m1 = net64._processor.gnn_stacks
    f64 = dict(y=y6, dx=xg6.grad, de=eg6.grad, dW_edge_first0=m0.edge_fn[0][0].weight.grad, dW_edge_hidden0=m0.edge_fn[0][2].weight.grad,
               dW_edge_last1=m1.edge_fn[0][4].weight.grad, db_edge_last1=m1.edge_fn[0][4].bias.grad,
               dW_node_first1=m1.node_fn[0][0].weight.grad, dgamma_edge0=m0.edge_fn[1].weight.grad, dbeta_node1=m1.node_fn[1].bias.grad,
               dW_enc_edge=net64._encoder.edge_fn[0][0].weight.grad, dW_dec_last=net64._decoder.node_fn[4].weight.grad)
    for k, v in f64.items():
        ref32 = np.asarray(out[k], np.float64)
        out["f64dist." + k] = np.asarray(np.abs(npy(v) - ref32).max() / (np.abs(npy(v)).max() + 1e-30))
    np.savez_compressed(os.path.join(OUT, "gnn128.npz"), **out)


def gen_sim128():
    """The two simulator wrappers of SURVEY a9 at the config-4 WIDTH (latent 128, hidden 128: the width at which the build's HIP kernels
    -- csplat_linear128 / csplat_gnn_node_update under no_grad, the fused autograd nodes in training -- take over; the reference-run
    fixtures meshsim.npz / gnn.npz are at latent 32 and exercise the generic path only): the reference's own
    ClothMeshSimulator (meshnet/cloth_network.py:13-254, normalize=True) and MeshSimulator (meshnet/meshnet_network.py:14-191, its text
    exec'd with the merge conflict resolved to 9b63d7a) under the PyG shim, weights in closed form (not stored), on a graph of E = 17,400
    edges (>= the row count from which the build's training path runs its fused nodes): train-mode call with noise (online normaliser
    statistics), a decoder / encoder weight gradient, eval-mode prediction."""
    install_pyg_shim()
    from meshnet.cloth_network import ClothMeshSimulator
    viz = types.ModuleType("meshnet.viz")
    viz.plot_mesh = viz.plot_pcd_list = None
    sys.modules["meshnet.viz"] = viz
    src = open(os.path.join(REF, "meshnet/meshnet_network.py")).read()
    src = re.sub(r"<<<<<<< HEAD\n.*?=======\n(.*?)>>>>>>> [^\n]*\n", r"\1", src, flags=re.S)
    ns = {"__name__": "meshnet.meshnet_network"}
    exec(compile(src, "<meshnet_network>", "exec"), ns)
    MeshSimulator = ns["MeshSimulator"]
    g = torch.Generator().manual_seed(1280)
    N, deg = 580, 30
    E = N * deg
    dst = torch.arange(N).repeat_interleave(deg)
    src_ = (dst + torch.randint(1, 45, (E,), generator=g)) % N
    perm = torch.randperm(E, generator=g)
    ei = torch.stack([src_, dst])[:, perm].contiguous()
    ef = torch.randn(E, 4, generator=g)
    out = dict(edge_index=npy(ei).astype(np.int32), edge_features=npy(ef))
    # ---- ClothMeshSimulator
    sim = closed_form_weights(ClothMeshSimulator(simulation_dimensions=3, nnode_in=8, nedge_in=4, latent_dim=128, nmessage_passing_steps=3,
                                                 nmlp_layers=2, mlp_hidden_dim=128, nnode_types=2, node_type_embedding_size=2,
                                                 normalize=True, device="cpu"), salt=3)
    vel = torch.randn(N, 6, generator=g) * 0.1
    ntype = torch.randint(0, 2, (N, 1), generator=g)
    tgt = torch.randn(N, 3, generator=g) * 0.1
    noise = torch.randn(N, 6, generator=g) * 0.01
    w = torch.randn(N, 3, generator=g)
    sim.train()
    pa, ta = sim.predict_acceleration(vel, ntype, ei, ef, target_velocities=tgt, velocity_noise=noise)
    sim.zero_grad()
    pa2, _ = sim.predict_acceleration(vel, ntype, ei, ef, target_velocities=tgt, velocity_noise=noise)
    (pa2 * w).sum().backward()
    epd = sim._encode_process_decode
    out.update(c_vel=npy(vel), c_type=npy(ntype), c_tgt=npy(tgt), c_noise=npy(noise), c_w=npy(w), c_pred_acc=npy(pa), c_tgt_acc=npy(ta),
               c_pred_acc2=npy(pa2), c_dW_dec=npy(epd._decoder.node_fn[0].weight.grad),
               c_dW_edge_hidden1=npy(epd._processor.gnn_stacks[1].edge_fn[0][2].weight.grad),
               c_w_probe=npy(epd._processor.gnn_stacks[0].edge_fn[0][2].weight[:2, :5]))
    for nm in ("_output_normalizer", "_node_normalizer"):
        for k, v in getattr(sim, nm).get_variable().items():
            if torch.is_tensor(v):
                out[f"c{nm}.{k}"] = npy(v)
    sim.eval()
    out["c_pred_vel"] = npy(sim.predict_velocity(vel, ntype, ei, ef))
    # ---- MeshSimulator
    ms = closed_form_weights(MeshSimulator(simulation_dimensions=3, nnode_in=6, nedge_in=4, latent_dim=128, nmessage_passing_steps=3,
                                           nmlp_layers=2, mlp_hidden_dim=128, nnode_types=2, node_type_embedding_size=2, device="cpu"), salt=5)
    pos = torch.randn(N, 3, generator=g)
    # (a per-node time: with ONE time value for all 580 nodes the reference's Normalizer takes sqrt(E[t^2] - E[t]^2) of a column whose
    #  variance is zero up to fp32 rounding -- it came out negative in the reference's own run and the prediction was NaN)
    tvec = 0.35 + 0.1 * torch.rand(N, generator=g)
    mtgt = pos + torch.randn(N, 3, generator=g) * 0.05
    mnoise = torch.randn(N, 3, generator=g) * 0.01
    ms.train()
    pd1, td1 = ms.predict_dx(pos, tvec, ntype, ei, ef, target_positions=mtgt, position_noise=mnoise)
    ms.zero_grad()
    pd2, _ = ms.predict_dx(pos, tvec, ntype, ei, ef, target_positions=mtgt, position_noise=mnoise)
    (pd2 * w).sum().backward()
    mepd = ms._encode_process_decode
    out.update(m_pos=npy(pos), m_time=npy(tvec), m_tgt=npy(mtgt), m_noise=npy(mnoise), m_dx1_pred=npy(pd1), m_dx1_target=npy(td1),
               m_dx2_pred=npy(pd2), m_dW_dec=npy(mepd._decoder.node_fn[0].weight.grad),
               m_dW_enc=npy(mepd._encoder.node_fn[0][0].weight.grad))
    for nm in ("_output_normalizer", "_node_normalizer"):
        for k, v in getattr(ms, nm).get_variable().items():
            if torch.is_tensor(v):
                out[f"m{nm}.{k}"] = npy(v)
    ms.eval()
    out["m_position"] = npy(ms.predict_position(pos, tvec[:, None], ntype, ei, ef))
    np.savez_compressed(os.path.join(OUT, "sim128.npz"), **out)


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
    # round 5: the reference module's own PARAMETER GRADIENTS (autograd over its three nn.Linear layers) for a fixed cotangent per time --
    # what csplat_sim_hidden_fwd/_bwd + csplat_rows_dot_fwd/_bwd are held to on the GPU (tests/test_reference_goldens_gpu.py)
    gw = torch.Generator().manual_seed(61)
    w = torch.randn(len(times), V, 3, generator=gw)
    sim.zero_grad()
    loss = sum((sim(torch.tensor(tt, dtype=torch.float32).repeat(V, 1)) * w[i]).sum() for i, tt in enumerate(times))
    loss.backward()
    out["res_grad_w"] = npy(w)
    out["res_loss"] = npy(loss)
    for k, prm in sim.named_parameters():
        if prm.grad is not None:
            out["res_grad." + k] = npy(prm.grad)
    np.savez(os.path.join(OUT, "simulator.npz"), **out)


def install_mesh_shims():
    """the empty stand-ins gen_densify describes, so that scene_reconstruction.gaussian_mesh imports on this image"""
    install_pyg_shim()
    sys.modules["torch_geometric"].utils = types.ModuleType("torch_geometric.utils")
    sys.modules["torch_geometric.utils"] = sys.modules["torch_geometric"].utils
    for name in ("h5py", "roma", "plyfile", "simple_knn", "simple_knn._C"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    src = open(os.path.join(REF, "meshnet", "data_utils.py")).read()
    m = re.search(r"^def compute_barycentric_coordinates\(.*?(?=^\S)", src, re.S | re.M)
    du = types.ModuleType("meshnet.data_utils")
    exec(compile("import torch\n" + m.group(0), "data_utils.py:compute_barycentric_coordinates", "exec"), du.__dict__)
    for nm in ("compute_mesh", "compute_edge_features", "load_mesh_from_h5py", "vertice_rotation"):
        setattr(du, nm, None)
    import meshnet
    sys.modules["meshnet.data_utils"] = du
    meshnet.data_utils = du


class cuda_as_cpu:
    """`device="cuda"` in the reference's factory calls lands on the CPU while the block runs"""

    def __enter__(self):
        def cpu_factory(fn):
            def w(*a, **k):
                if k.get("device") is not None and "cuda" in str(k["device"]):
                    k["device"] = "cpu"
                return fn(*a, **k)
            return w
        self.saved = {n: getattr(torch, n) for n in ("zeros", "ones", "arange", "empty", "tensor", "normal")}
        for n, f in self.saved.items():
            setattr(torch, n, cpu_factory(f))
        torch.cuda.empty_cache = lambda: None
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)


def gen_densify():
    """scene_reconstruction.gaussian_mesh.MultiGaussianMesh densify / prune / opacity reset + the Adam-state surgery of
    gaussian_model.py:266-341, run HERE on CPU tensors: the module's hard imports that are not installed (h5py, roma,
    plyfile, simple_knn, torch_geometric, meshnet.data_utils' plotting deps) are replaced by empty shim modules -- none of
    their symbols is reached by the methods exercised -- and `device="cuda"` in the factory calls is mapped to the CPU.
    compute_barycentric_coordinates is taken from meshnet/data_utils.py by exec'ing that one function's source."""
    install_mesh_shims()
    with cuda_as_cpu():
        from scene_reconstruction.gaussian_mesh import MultiGaussianMesh
        g = torch.Generator().manual_seed(77)
        V, F, P = 12, 14, 60
        pos = torch.rand(V, 3, generator=g)
        face = torch.stack([torch.randperm(V, generator=g)[:3] for _ in range(F)], 1)          # [3, F]
        pc = MultiGaussianMesh(3)
        pc.mesh = types.SimpleNamespace(pos=pos, face=face)
        pc.face_ids = torch.randint(0, F, (P,), generator=g)
        mk = lambda *s, scale=1.0: torch.nn.Parameter((torch.randn(*s, generator=g) * scale))  # noqa: E731
        bary = torch.rand(P, 3, generator=g) + 0.05
        pc.face_bary = torch.nn.Parameter(bary / bary.sum(1, keepdim=True))
        pc.face_offset = mk(P, 1, scale=0.01)
        pc._features_dc, pc._features_rest = mk(P, 1, 3), mk(P, 15, 3, scale=0.1)
        pc._opacity = mk(P, 1, scale=2.0)
        pc._scaling = torch.nn.Parameter(torch.log(torch.rand(P, 3, generator=g) * 0.05 + 0.002))
        pc._rotation = mk(P, 4)
        pc.percent_dense = 0.01
        pc.max_radii2D = torch.rand(P, generator=g) * 40.0
        pc.pos_gradient_accum, pc.denom = torch.zeros(P, 1), torch.zeros(P, 1)
        names = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
        params = [pc.face_bary, pc.face_offset, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation]
        lrs = [1.6e-4, 1.6e-4, 2.5e-3, 2.5e-3 / 20, 0.05, 0.005, 0.001]
        pc.optimizer = torch.optim.Adam([{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(params, lrs, names)], lr=0.0,
                                        eps=1e-15)
        out = {"pos": pos, "face": face, "face_ids": pc.face_ids, "max_radii2D": pc.max_radii2D}
        for n, p_ in zip(names, params):
            out["init." + n] = p_.detach().clone()
        grads = []
        for it in range(3):                                    # three Adam steps so that the state is populated
            for p_ in params:
                p_.grad = torch.randn(p_.shape, generator=g) * 0.1
                grads.append(p_.grad.clone())
            pc.optimizer.step()
        out["adam_grads"] = torch.cat([x.reshape(-1) for x in grads])

        def dump(tag):
            for grp in pc.optimizer.param_groups:
                p_ = grp["params"][0]
                st = pc.optimizer.state[p_]
                out[f"{tag}.{grp['name']}"] = p_.detach().clone()
                out[f"{tag}.{grp['name']}.exp_avg"] = st["exp_avg"].clone()
                out[f"{tag}.{grp['name']}.exp_avg_sq"] = st["exp_avg_sq"].clone()
                out[f"{tag}.{grp['name']}.step"] = torch.as_tensor(float(st["step"]))
            out[f"{tag}.face_ids"] = pc.face_ids.clone()
            out[f"{tag}.pos_gradient_accum"], out[f"{tag}.denom"] = pc.pos_gradient_accum.clone(), pc.denom.clone()
            out[f"{tag}.max_radii2D"] = pc.max_radii2D.clone()
        dump("stepped")
        vsp = torch.randn(P, 3, generator=g) * 2e-4
        upd = torch.rand(P, generator=g) > 0.3
        out["vsp"], out["update_filter"] = vsp, upd
        pc.add_densification_stats(vsp, upd)
        pc.add_densification_stats(vsp * 0.5, upd)
        dump("stats")
        torch.manual_seed(4321)                                  # densify_and_split draws torch.normal from the global RNG
        pc.densify(2e-4, 0.05, 1.0, None)
        dump("densified")
        pc.prune(2e-4, 0.3, 1.0, 20)
        dump("pruned")
        pc.reset_opacity()
        dump("reset")
        # one more Adam step after all the surgery: the state must still be consistent with the parameters
        post = []
        for grp in pc.optimizer.param_groups:
            p_ = grp["params"][0]
            p_.grad = torch.randn(p_.shape, generator=g) * 0.1
            post.append(p_.grad.clone())
        pc.optimizer.step()
        out["post_grads"] = torch.cat([x.reshape(-1) for x in post])
        dump("after_step")
        # ---- cleanup_barycentric_coordinates (gaussian_mesh.py:267-322) on a small grid mesh: Gaussians whose barycentric
        # coordinates went negative hop to the face across the offending edge (or are pushed back on a border edge)
        gm = 6
        xs = torch.linspace(0, 1, gm)
        gpos = torch.stack([xs.repeat(gm), xs.repeat_interleave(gm), 0.05 * torch.rand(gm * gm, generator=g)], 1)
        quads = [(r * gm + c, r * gm + c + 1, (r + 1) * gm + c, (r + 1) * gm + c + 1) for r in range(gm - 1) for c in range(gm - 1)]
        gface = torch.tensor([[a, b, c2] for a, b, c2, d in quads] + [[b, d, c2] for a, b, c2, d in quads]).t().contiguous()
        Pc = 90
        pc2 = MultiGaussianMesh(3)
        pc2.mesh = types.SimpleNamespace(pos=gpos, face=gface)
        pc2.face_ids = torch.randint(0, gface.shape[1], (Pc,), generator=g)
        b = torch.rand(Pc, 3, generator=g)
        b = b / b.sum(1, keepdim=True)
        neg = torch.rand(Pc, 3, generator=g) < 0.22                      # some rows get one, two or three negatives
        b = torch.where(neg, -0.3 * torch.rand(Pc, 3, generator=g) - 0.01, b)
        pc2.face_bary = torch.nn.Parameter(b.clone())
        out["cleanup.pos"], out["cleanup.face"] = gpos, gface
        out["cleanup.face_ids_in"], out["cleanup.face_bary_in"] = pc2.face_ids.clone(), b.clone()
        with torch.no_grad():
            pc2.cleanup_barycentric_coordinates()
        out["cleanup.face_ids_out"], out["cleanup.face_bary_out"] = pc2.face_ids.clone(), pc2.face_bary.detach().clone()
        print("cleanup: rows with a negative coordinate", int((b < 0).any(1).sum()), "faces changed",
              int((out["cleanup.face_ids_out"] != out["cleanup.face_ids_in"]).sum()))
    np.savez_compressed(os.path.join(OUT, "densify.npz"), **{k: npy(v) if torch.is_tensor(v) else v for k, v in out.items()})
    print("densify.npz: P", P, "->", int(out["densified.face_ids"].shape[0]), "->", int(out["pruned.face_ids"].shape[0]))


def gen_scene_io():
    """scene_reconstruction.dataset_readers.readCamerasFromTransforms / read_timeline on a tiny Blender-style scene that this
    function WRITES under tests/golden/blender_scene/ (2 views x 3 times, 8x6 RGBA PNGs, a gripper mask for one frame): the
    fixture is the scene files + what the reference's reader returns for them.  The module's hard imports that are not
    installed (h5py, plyfile, torchvision, simple_knn, ...) are replaced by empty modules -- none is reached by the two
    functions exercised."""
    import json
    from PIL import Image
    for name in ("h5py", "plyfile", "torchvision", "torchvision.transforms", "simple_knn", "simple_knn._C", "roma", "open3d"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["plyfile"], "PlyData"):
        sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    if not hasattr(sys.modules["simple_knn._C"], "distCUDA2"):
        sys.modules["simple_knn._C"].distCUDA2 = None
    install_pyg_shim()
    import scene_reconstruction.dataset_readers as dr
    # dataset_readers.py:371 builds the composited image with Image.fromarray(np.array(arr * 255.0, dtype=np.byte), "RGB").
    # Pillow < 10 took the int8 buffer as the 8-bit RGB bytes it is; the Pillow installed here (12) rejects the dtype.  The
    # old behaviour is restored for the call ("shim-derived", like the PyG shim): same bytes, viewed as uint8.
    real_fromarray = Image.fromarray

    def fromarray_compat(obj, mode=None):
        if isinstance(obj, np.ndarray) and obj.dtype == np.int8:
            obj = obj.view(np.uint8)
        return real_fromarray(obj, mode) if mode is not None and obj.dtype != np.uint8 else real_fromarray(obj)
    dr.Image.fromarray = fromarray_compat
    root = os.path.join(OUT, "blender_scene")
    rng = np.random.default_rng(11)
    W, H = 8, 6
    views = []
    for v in range(2):      # two camera-to-world matrices: rotation about z and x, translated
        az, el = 0.7 + 1.1 * v, -0.4 + 0.3 * v
        Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(el), -np.sin(el)], [0, np.sin(el), np.cos(el)]])
        c2w = np.eye(4)
        c2w[:3, :3] = Rz @ Rx
        c2w[:3, 3] = [1.5 - v, 0.3 * v, 2.0 + 0.5 * v]
        views.append(c2w)
    for split, times, named in (("train", [0.0, 0.5, 1.0], True), ("test", [0.25, 1.0], False)):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for vi, c2w in enumerate(views):
            for ti, t in enumerate(times):
                name = f"r_{vi}_{ti}" if named else f"img{vi}{ti}"      # the second form exercises the unique-transform ids
                img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
                img[..., 3] = rng.choice([0, 128, 255], (H, W))
                Image.fromarray(img, "RGBA").save(os.path.join(root, split, name + ".png"))
                frames.append({"file_path": f"./{split}/{name}", "time": t, "transform_matrix": c2w.tolist()})
        json.dump({"camera_angle_x": 0.6911, "camera_angle_y": 0.5273, "frames": frames},
                  open(os.path.join(root, f"transforms_{split}.json"), "w"), indent=1)
    os.makedirs(os.path.join(root, "masks_gripper"), exist_ok=True)
    for vi in range(2):
        for ti in range(3):
            Image.fromarray((rng.random((H, W)) > 0.5).astype(np.uint8) * 255, "L").save(
                os.path.join(root, "masks_gripper", f"r_{vi}_{ti}.png"))
    out = {}
    mapper, max_time = dr.read_timeline(root)
    out["timeline.keys"] = np.array(sorted(mapper)); out["timeline.values"] = np.array([mapper[k] for k in sorted(mapper)])
    out["timeline.max"] = np.array(max_time)
    cases = (("train_white", "transforms_train.json", True, None, None), ("train_black_skip", "transforms_train.json", False, 2, 2),
             ("test_white", "transforms_test.json", True, None, None))
    import shutil
    for tag, tf, white, tskip, vskip in cases:
        if tag.startswith("test"):       # the mask directory only covers the train names
            shutil.move(os.path.join(root, "masks_gripper"), os.path.join(root, "_masks_off"))
        try:
            infos = dr.readCamerasFromTransforms(root, tf, white, ".png", mapper, time_skip=tskip, view_skip=vskip)
        finally:
            if tag.startswith("test"):
                shutil.move(os.path.join(root, "_masks_off"), os.path.join(root, "masks_gripper"))
        out[f"{tag}.n"] = np.array(len(infos))
        for i, c in enumerate(infos):
            out[f"{tag}.{i}.R"], out[f"{tag}.{i}.T"] = np.asarray(c.R), np.asarray(c.T)
            out[f"{tag}.{i}.fov"] = np.array([c.FovX, c.FovY])
            out[f"{tag}.{i}.image"] = npy(c.image)
            out[f"{tag}.{i}.ints"] = np.array([c.uid, c.width, c.height, int(c.view_id), int(c.time_id)])
            out[f"{tag}.{i}.time"] = np.array(c.time)
            out[f"{tag}.{i}.name"] = np.array(c.image_name)
            if c.mask is not None:
                out[f"{tag}.{i}.mask"] = npy(c.mask)
    dr.Image.fromarray = real_fromarray
    np.savez_compressed(os.path.join(OUT, "scene_io.npz"), **out)
    print("scene_io.npz:", {t: int(out[f"{t}.n"]) for t, *_ in cases})


def gen_meshsim():
    """meshnet.meshnet_network.MeshSimulator (:14-191): predict_dx in train mode (noise, online normaliser statistics, twice so
    that the second call normalises with accumulated statistics) and predict_position in eval mode, under the PyG shim.  The
    file does not parse as shipped (merge-conflict markers): its text is exec'd with the conflict resolved to the
    `9b63d7a` side and `meshnet.viz` (matplotlib / imageio plotting helpers, not reached) replaced by an empty module."""
    install_pyg_shim()
    viz = types.ModuleType("meshnet.viz")
    viz.plot_mesh = viz.plot_pcd_list = None
    sys.modules["meshnet.viz"] = viz
    src = open(os.path.join(REF, "meshnet/meshnet_network.py")).read()
    src = re.sub(r"<<<<<<< HEAD\n.*?=======\n(.*?)>>>>>>> [^\n]*\n", r"\1", src, flags=re.S)
    ns = {"__name__": "meshnet.meshnet_network"}
    exec(compile(src, "<meshnet_network>", "exec"), ns)
    MeshSimulator = ns["MeshSimulator"]
    g = torch.Generator().manual_seed(21)
    N, E = 40, 220
    ei = torch.randint(0, N, (2, E), generator=g)
    ef = torch.randn(E, 4, generator=g)
    torch.manual_seed(11)
    # node features: 3 position + 1 time + 2 one-hot
    sim = MeshSimulator(simulation_dimensions=3, nnode_in=6, nedge_in=4, latent_dim=32, nmessage_passing_steps=2,
                        nmlp_layers=2, mlp_hidden_dim=32, nnode_types=2, node_type_embedding_size=2, device="cpu")
    pos = torch.randn(N, 3, generator=g)
    tvec = torch.full((N,), 0.35)                   # 1-D time vector: exercises the [:, None] branch (:97-98)
    ntype = torch.randint(0, 2, (N, 1), generator=g)
    tgt = pos + torch.randn(N, 3, generator=g) * 0.05
    noise = torch.randn(N, 3, generator=g) * 0.01
    out = dict(edge_index=npy(ei), edge_features=npy(ef), pos=npy(pos), time=npy(tvec), node_type=npy(ntype), target=npy(tgt),
               noise=npy(noise))
    for k, v in sim.state_dict().items():
        out["sd." + k] = npy(v)
    sim.train()
    pd1, td1 = sim.predict_dx(pos, tvec, ntype, ei, ef, target_positions=tgt, position_noise=noise)
    pos2 = pos * 1.3 + 0.2
    tvec2 = torch.full((N, 1), 0.7)
    pd2, td2 = sim.predict_dx(pos2, tvec2, ntype, ei, ef, target_positions=tgt, position_noise=noise)
    out.update(dx1_pred=npy(pd1), dx1_target=npy(td1), pos2=npy(pos2), time2=npy(tvec2), dx2_pred=npy(pd2), dx2_target=npy(td2))
    # gradient of sum(pred * w) + sum(target_norm * w2) w.r.t. one decoder and one encoder weight
    w = torch.randn(N, 3, generator=g)
    sim.zero_grad()
    pd3, _ = sim.predict_dx(pos, tvec, ntype, ei, ef, target_positions=tgt, position_noise=noise)
    (pd3 * w).sum().backward()
    out.update(dx3_w=npy(w), dx3_pred=npy(pd3),
               dx3_dW_dec=npy(sim._encode_process_decode._decoder.node_fn[0].weight.grad),
               dx3_dW_enc=npy(sim._encode_process_decode._encoder.node_fn[0][0].weight.grad))
    for nm in ("_output_normalizer", "_node_normalizer"):
        for k, v in getattr(sim, nm).get_variable().items():
            if torch.is_tensor(v):
                out[f"{nm}.{k}"] = npy(v)
    sim.eval()
    pp = sim.predict_position(pos, tvec[:, None], ntype, ei, ef)
    pd_eval, none = sim.predict_dx(pos, tvec, ntype, ei, ef)
    assert none is None
    out.update(position=npy(pp), dx_eval=npy(pd_eval))
    np.savez_compressed(os.path.join(OUT, "meshsim.npz"), **out)


def install_roma_scipy():
    """`roma` is not installed and cannot be (no network).  For the get_rotation fixture its three entry points are served
    by SCIPY (an independent, third-party implementation of the same published operations), in float64:
      rigid_points_registration(x, y) -> (R, t) minimising |R x + t - y|   == Rotation.align_vectors on the centred sets
      rotmat_to_unitquat(R)           -> xyzw                              == Rotation.from_matrix(R).as_quat()
      quat_composition([p, q])        -> Hamilton product p*q, xyzw        == (Rotation.from_quat(p) * from_quat(q)).as_quat()
    The fixture is therefore "scipy-derived": it pins the CALL PATTERN of gaussian_mesh.py:171-188 (which quaternion goes
    where, in which convention -- SURVEY F8) and the Kabsch solution against code the build did not write."""
    from scipy.spatial.transform import Rotation
    roma = sys.modules.get("roma") or types.ModuleType("roma")

    def rigid_points_registration(x, y, weights=None, compute_scaling=False):
        xs, ys = x.detach().double().numpy(), y.detach().double().numpy()
        Rs, ts = [], []
        for xi, yi in zip(xs, ys):
            cx, cy = xi.mean(0), yi.mean(0)
            R, _ = Rotation.align_vectors(yi - cy, xi - cx)
            Rs.append(R.as_matrix()); ts.append(cy - R.as_matrix() @ cx)
        return torch.tensor(np.stack(Rs)), torch.tensor(np.stack(ts))

    def rotmat_to_unitquat(R):
        return torch.tensor(Rotation.from_matrix(R.detach().double().numpy()).as_quat())

    def quat_composition(seq, normalize=False):
        acc = Rotation.from_quat(seq[0].detach().double().numpy())
        for q in seq[1:]:
            acc = acc * Rotation.from_quat(q.detach().double().numpy())
        return torch.tensor(acc.as_quat())
    roma.rigid_points_registration, roma.rotmat_to_unitquat, roma.quat_composition = \
        rigid_points_registration, rotmat_to_unitquat, quat_composition
    sys.modules["roma"] = roma


def gen_mesh_transform():
    """MultiGaussianMesh.get_xyz (gaussian_mesh.py:151-169: rest pose and deformed, forward + autograd gradients -- pure torch,
    the reference's own arithmetic) and get_rotation (:171-188, with roma served by scipy, see install_roma_scipy; its vertex
    gradient is pinned by central differences of that float64 forward)."""
    install_mesh_shims()
    install_roma_scipy()
    with cuda_as_cpu():
        from scene_reconstruction.gaussian_mesh import MultiGaussianMesh
        g = torch.Generator().manual_seed(31)
        gm = 7
        xs = torch.linspace(-0.5, 0.5, gm)
        pos = torch.stack([xs.repeat(gm), xs.repeat_interleave(gm), 0.05 * torch.rand(gm * gm, generator=g)], 1)
        quads = [(r * gm + c, r * gm + c + 1, (r + 1) * gm + c, (r + 1) * gm + c + 1) for r in range(gm - 1) for c in range(gm - 1)]
        face = torch.tensor([[a, b, c2] for a, b, c2, d in quads] + [[b, d, c2] for a, b, c2, d in quads]).t().contiguous()
        V, F, P = pos.shape[0], face.shape[1], 150
        pc = MultiGaussianMesh(3)
        pc.mesh = types.SimpleNamespace(pos=pos, face=face)
        pc.face_ids = torch.randint(0, F, (P,), generator=g)
        bary = torch.rand(P, 3, generator=g) + 0.02
        bary = bary / bary.sum(1, keepdim=True) * (0.8 + 0.4 * torch.rand(P, 1, generator=g))   # rows that do NOT sum to 1
        pc.face_bary = torch.nn.Parameter(bary.clone())
        pc._rotation = torch.nn.Parameter(torch.randn(P, 4, generator=g))
        # a rigid motion + a smooth wave + noise: large rotations incl. > 90 degrees for some faces
        ang = 2.1
        Rz = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
        Rx = torch.tensor([[1, 0, 0], [0, np.cos(0.9), -np.sin(0.9)], [0, np.sin(0.9), np.cos(0.9)]], dtype=torch.float32)
        deformed = (pos @ (Rz @ Rx).T) + torch.tensor([0.1, -0.2, 0.3]) + 0.03 * torch.randn(V, 3, generator=g)
        deformed[:, 2] += 0.1 * torch.sin(4 * pos[:, 0])
        out = dict(pos=pos, face=face, face_ids=pc.face_ids, face_bary=bary, rotation=pc._rotation.detach().clone(),
                   deformed=deformed)
        out["xyz_rest"] = pc.get_xyz().detach()
        dv = deformed.clone().requires_grad_(True)
        xyz = pc.get_xyz(dv)
        wx = torch.randn(P, 3, generator=g)
        (xyz * wx).sum().backward()
        out.update(xyz_deformed=xyz.detach(), xyz_w=wx, xyz_d_vertices=dv.grad.clone(), xyz_d_bary=pc.face_bary.grad.clone())
        out["rot_rest"] = pc.get_rotation().detach()
        q0 = pc.get_rotation(deformed).detach()
        out["rot_deformed"] = q0                                             # float64, defined up to the sign of each row
        wq = torch.randn(P, 4, generator=g).double()
        out["rot_w"] = wq

        def f(dvert):
            q = pc.get_rotation(dvert).detach()
            sgn = torch.sign((q * q0).sum(1, keepdim=True))
            return float(((q * sgn) * wq).sum())
        h = 1e-4
        gnum = torch.zeros(V, 3, dtype=torch.float64)
        for i in range(V):
            for c in range(3):
                dp, dm = deformed.double().clone(), deformed.double().clone()
                dp[i, c] += h; dm[i, c] -= h
                gnum[i, c] = (f(dp) - f(dm)) / (2 * h)
        out["rot_d_vertices_fd"] = gnum
    np.savez_compressed(os.path.join(OUT, "mesh_transform.npz"), **{k: npy(v) for k, v in out.items()})


def gen_vertice_rotation():
    """MultiGaussianMesh.get_vertice_rotation (gaussian_mesh.py:190-201) run here: `vertice_rotation` / `axis_angle_to_quat` are the
    reference's own text (exec'd from meshnet/data_utils.py:460-491); torch_geometric's GenerateMeshNormals is absent and served by a
    stand-in with PyG's published semantics (unit face normals summed onto the vertices, normalised) -- "shim-derived" for the normals,
    the reference's arithmetic for the quaternion and the call pattern (rest normals first)."""
    install_mesh_shims()
    import torch.nn.functional as F
    src = open(os.path.join(REF, "meshnet", "data_utils.py")).read()
    ns = {}
    for fn in ("axis_angle_to_quat", "vertice_rotation"):
        m = re.search(r"^def %s\(.*?(?=^\S)" % fn, src, re.S | re.M)
        exec(compile("import torch\n" + m.group(0), "data_utils.py:" + fn, "exec"), ns)

    class GenerateMeshNormals:
        def __call__(self, data):
            pos, face = data.pos, data.face
            fn = F.normalize(torch.cross(pos[face[1]] - pos[face[0]], pos[face[2]] - pos[face[0]], dim=1), p=2, dim=-1)
            idx = torch.cat([face[0], face[1], face[2]], dim=0)
            norm = torch.zeros_like(pos).index_add_(0, idx, fn.repeat(3, 1))
            data.norm = F.normalize(norm, p=2, dim=-1)
            return data
    tg = sys.modules["torch_geometric"]
    tg.transforms = types.ModuleType("torch_geometric.transforms")
    tg.transforms.GenerateMeshNormals = GenerateMeshNormals
    sys.modules["torch_geometric.transforms"] = tg.transforms
    with cuda_as_cpu():
        import scene_reconstruction.gaussian_mesh as gmod
        gmod.vertice_rotation = ns["vertice_rotation"]
        g = torch.Generator().manual_seed(17)
        gm = 6
        xs = torch.linspace(-0.5, 0.5, gm)
        pos = torch.stack([xs.repeat(gm), xs.repeat_interleave(gm), 0.08 * torch.rand(gm * gm, generator=g)], 1)
        quads = [(r * gm + c, r * gm + c + 1, (r + 1) * gm + c, (r + 1) * gm + c + 1) for r in range(gm - 1) for c in range(gm - 1)]
        face = torch.tensor([[a, b, c2] for a, b, c2, d in quads] + [[b, d, c2] for a, b, c2, d in quads]).t().contiguous()
        pc = gmod.MultiGaussianMesh(3)
        rest = GenerateMeshNormals()(types.SimpleNamespace(pos=pos, face=face)).norm       # what compute_mesh leaves in mesh.norm
        pc.mesh = types.SimpleNamespace(pos=pos, face=face, norm=rest)
        ang = 0.7
        Rx = torch.tensor([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]], dtype=torch.float32)
        deformed = pos @ Rx.T + 0.02 * torch.randn(pos.shape[0], 3, generator=g)
        deformed[:, 2] += 0.15 * torch.sin(5 * pos[:, 0])
        q = pc.get_vertice_rotation(deformed)
    np.savez_compressed(os.path.join(OUT, "vertice_rotation.npz"), pos=npy(pos), face=npy(face), deformed=npy(deformed), rest_norm=npy(rest),
                        quat=npy(q))


class RecordingRasterizer:
    """stand-in for the absent `diff_gaussian_rasterization` extension: records the settings record and every tensor the
    reference's render() hands the rasterizer, returns zeros of the documented shapes (color [3,H,W], radii [P] int32, depth [1,H,W])"""
    calls = []

    def __init__(self, raster_settings):
        self.raster_settings = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        RecordingRasterizer.calls.append((self.raster_settings, dict(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs,
                                                                     colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                                                                     cov3D_precomp=cov3D_precomp)))
        rs = self.raster_settings
        P = means3D.shape[0]
        return (torch.zeros(3, rs.image_height, rs.image_width), torch.zeros(P, dtype=torch.int32),
                torch.zeros(1, rs.image_height, rs.image_width))


def install_render_shims():
    """what `import gaussian_renderer` (the reference's) needs on this image: the mesh shims + roma-by-scipy of the fixtures above,
    a recording `diff_gaussian_rasterization`, and `meshnet.meshnet_network` exec'd from the reference text with its merge
    conflict resolved to the 9b63d7a side (as gen_meshsim does)."""
    from typing import NamedTuple
    install_mesh_shims()
    install_roma_scipy()

    class GaussianRasterizationSettings(NamedTuple):       # the 12 fields of the upstream record, in upstream's order
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool
    dgr = types.ModuleType("diff_gaussian_rasterization")
    dgr.GaussianRasterizationSettings, dgr.GaussianRasterizer = GaussianRasterizationSettings, RecordingRasterizer
    sys.modules["diff_gaussian_rasterization"] = dgr
    viz = types.ModuleType("meshnet.viz")
    viz.plot_mesh = viz.plot_pcd_list = None
    sys.modules["meshnet.viz"] = viz
    src = open(os.path.join(REF, "meshnet/meshnet_network.py")).read()
    src = re.sub(r"<<<<<<< HEAD\n.*?=======\n(.*?)>>>>>>> [^\n]*\n", r"\1", src, flags=re.S)
    mm = types.ModuleType("meshnet.meshnet_network")
    exec(compile(src, "<meshnet_network>", "exec"), mm.__dict__)
    import meshnet
    sys.modules["meshnet.meshnet_network"] = mm
    meshnet.meshnet_network = mm


class cuda_calls_as_cpu(cuda_as_cpu):
    """cuda_as_cpu + `tensor.cuda()` -> the tensor itself and `zeros_like(..., device="cuda")` on the CPU (render() uses both)"""

    def __enter__(self):
        super().__enter__()
        self.saved_cuda, self.saved_zl = torch.Tensor.cuda, torch.zeros_like
        torch.Tensor.cuda = lambda t, *a, **k: t

        def zeros_like(x, **k):
            if "cuda" in str(k.get("device", "")):
                k["device"] = "cpu"
            return self.saved_zl(x, **k)
        torch.zeros_like = zeros_like
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.zeros_like = self.saved_cuda, self.saved_zl
        super().__exit__(*exc)


def render_wiring_scene():
    """the seeded inputs of render_wiring.npz as plain tensors (also imported by tests/test_render_wiring_cpu.py, which feeds the
    reference's own MultiGaussianMesh / Camera objects built from them to the BUILD's render())"""
    g = torch.Generator().manual_seed(77)
    gm = 6
    xs = torch.linspace(-0.5, 0.5, gm)
    pos = torch.stack([xs.repeat(gm), xs.repeat_interleave(gm), 0.05 * torch.rand(gm * gm, generator=g)], 1)
    quads = [(r * gm + c, r * gm + c + 1, (r + 1) * gm + c, (r + 1) * gm + c + 1) for r in range(gm - 1) for c in range(gm - 1)]
    face = torch.tensor([[a, b, c2] for a, b, c2, d in quads] + [[b, d, c2] for a, b, c2, d in quads]).t().contiguous()
    F, P = face.shape[1], 90
    bary = torch.rand(P, 3, generator=g) + 0.05
    d = dict(pos=pos, face=face, face_ids=torch.randint(0, F, (P,), generator=g), face_bary=bary / bary.sum(1, keepdim=True),
             rotation=torch.randn(P, 4, generator=g), scaling=torch.log(0.02 + 0.03 * torch.rand(P, 3, generator=g)),
             opacity=torch.randn(P, 1, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
             features_rest=0.1 * torch.randn(P, 15, 3, generator=g), override_color=torch.rand(P, 3, generator=g),
             wave=0.08 * torch.randn(pos.shape[0], 3, generator=g), bg=torch.tensor([1.0, 0.5, 0.25]))
    # camera: the pose_spherical-style look-at of the synthetic scenes, through the reference's Camera class
    th, phi, rad = np.deg2rad(35.0), np.deg2rad(-30.0), 4.0
    c2w = np.eye(4)
    c2w[2, 3] = rad
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]])
    rot_th = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ rot_th @ rot_phi @ c2w
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    d.update(cam_R=torch.tensor(np.transpose(w2c[:3, :3])), cam_T=torch.tensor(w2c[:3, 3]), cam_FoVx=torch.tensor(0.6911),
             cam_FoVy=torch.tensor(0.52), cam_W=torch.tensor(72), cam_H=torch.tensor(56), cam_time=torch.tensor(0.4))
    return d


def render_wiring_objects(d, MultiGaussianMesh, Camera):
    """the reference's own `pc` and camera objects from the seeded arrays, and a simulator stand-in (any callable taking
    time_vector [V,1] and returning [V,3], which is all render() asks of it)"""
    pc = MultiGaussianMesh(3)
    pc.mesh = types.SimpleNamespace(pos=d["pos"], face=d["face"])
    pc.face_ids, pc.face_bary = d["face_ids"], torch.nn.Parameter(d["face_bary"].clone())
    pc._rotation, pc._scaling = torch.nn.Parameter(d["rotation"].clone()), torch.nn.Parameter(d["scaling"].clone())
    pc._opacity = torch.nn.Parameter(d["opacity"].clone())
    pc._features_dc, pc._features_rest = torch.nn.Parameter(d["features_dc"].clone()), torch.nn.Parameter(d["features_rest"].clone())
    pc.active_sh_degree = 2                                  # != max_sh_degree: settings.sh_degree must be the ACTIVE degree
    H, W = int(d["cam_H"]), int(d["cam_W"])
    cam = Camera(colmap_id=0, R=d["cam_R"].numpy(), T=d["cam_T"].numpy(), FoVx=float(d["cam_FoVx"]), FoVy=float(d["cam_FoVy"]),
                 image=torch.zeros(3, H, W), gt_alpha_mask=None, image_name="wiring", uid=0, data_device="cpu", time=float(d["cam_time"]))
    seen = []

    def simulator(time_vector):
        seen.append(time_vector.detach().clone())
        return d["pos"] + d["wave"] * time_vector            # [V,3] + [V,3] * [V,1]
    simulator.seen = seen
    return pc, cam, simulator


WIRING_CASES = {
    "default": dict(),
    "scale_mod": dict(scaling_modifier=1.7),
    "override_color": dict(override_color="override_color"),
    "static": dict(render_static=True),
    "project_vertices": dict(project_vertices=True),
    "cov_python": dict(pipe_cov=True),
}


def gen_render_wiring():
    """gaussian_renderer.render (:39-206), the reference's OWN function, run here with a recording stand-in for the rasterizer
    extension: what it puts into GaussianRasterizationSettings (tanfov = tan(FoV/2), bg, scale_modifier, the transposed matrices,
    sh_degree = ACTIVE degree, campos), which tensors it hands the rasterizer in which keyword (shs vs colors_precomp, scales vs
    cov3D_precomp, the deformed means / composed rotations, the zero screen-space tensor), what it asks of the simulator, and its
    by-products (projections, vertice_projections, the 14-field record).  roma is served by scipy (install_roma_scipy)."""
    install_render_shims()
    with cuda_calls_as_cpu():
        from gaussian_renderer import render
        from scene_reconstruction.cameras import Camera
        from scene_reconstruction.gaussian_mesh import MultiGaussianMesh
        d = render_wiring_scene()
        out = {"in." + k: npy(v) for k, v in d.items()}
        for name, kw in WIRING_CASES.items():
            pc, cam, sim = render_wiring_objects(d, MultiGaussianMesh, Camera)
            kw = dict(kw)
            pipe = types.SimpleNamespace(compute_cov3D_python=bool(kw.pop("pipe_cov", False)), convert_SHs_python=False, debug=False)
            if "override_color" in kw:
                kw["override_color"] = d["override_color"]
            RecordingRasterizer.calls.clear()
            res = render(cam, pc, sim, pipe, d["bg"], **kw)
            assert len(RecordingRasterizer.calls) == 1
            rs, args = RecordingRasterizer.calls[0]
            for f in rs._fields:
                v = getattr(rs, f)
                out[f"{name}.settings.{f}"] = npy(v) if torch.is_tensor(v) else np.asarray(v)
            for k, v in args.items():
                out[f"{name}.arg.{k}.none"] = np.asarray(v is None)
                if v is not None:
                    out[f"{name}.arg.{k}"] = npy(v)
            out[f"{name}.sim_calls"] = np.asarray(len(sim.seen))
            if sim.seen:
                out[f"{name}.sim_time_vector"] = npy(sim.seen[0])
            for f in res._fields:
                v = getattr(res, f)
                out[f"{name}.res.{f}.none"] = np.asarray(v is None)
                if v is not None:
                    out[f"{name}.res.{f}"] = npy(v)
            out[f"{name}.res.viewspace_points.requires_grad"] = np.asarray(res.viewspace_points.requires_grad)
    np.savez_compressed(os.path.join(OUT, "render_wiring.npz"), **out)


def gen_losses():
    """utils.loss_utils.l1_loss / ssim (:20-70; `lpips` -- imported at module level, never called here -- is an empty module)
    and scene_reconstruction.train_utils.image_losses / regularization (:50-102), whose two function bodies are exec'd from
    the reference text (the module itself imports wandb, imageio, lpips, the CUDA extensions ...).  Values and autograd
    gradients, masked and unmasked, for a [3,3,H,W] batch like the train step's."""
    sys.modules.setdefault("lpips", types.ModuleType("lpips"))
    from utils.loss_utils import l1_loss, ssim
    src = open(os.path.join(REF, "scene_reconstruction", "train_utils.py")).read()
    ns = {"torch": torch, "l1_loss": l1_loss, "ssim": ssim, "OptimizationParams": object}
    for fn in ("image_losses", "regularization"):
        m = re.search(r"^def %s\(.*?(?=^\S)" % fn, src, re.S | re.M)
        exec(compile(m.group(0), f"train_utils.py:{fn}", "exec"), ns)
    g = torch.Generator().manual_seed(41)
    B, H, W = 3, 37, 45                       # not multiples of the kernel's tile sizes
    gt = torch.rand(B, 3, H, W, generator=g)
    img = (gt + 0.15 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1.2)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.4).float()            # Camera.mask is [1,H,W] per view (cameras.py)
    out = dict(img=img, gt=gt, mask=mask)
    opt = types.SimpleNamespace(lambda_dssim=0.05, lambda_lpips=0)
    for tag, mk in (("plain", None), ("masked", mask)):
        x = img.clone().requires_grad_(True)
        l1 = l1_loss(x, gt, mk)
        l1.backward()
        out[f"{tag}.l1"], out[f"{tag}.l1_grad"] = l1.detach(), x.grad.clone()
        x = img.clone().requires_grad_(True)
        loss, d = ns["image_losses"](x, gt, opt, mk)
        loss.backward()
        out[f"{tag}.loss"], out[f"{tag}.loss_grad"] = loss.detach(), x.grad.clone()
        out[f"{tag}.ssim_loss"] = torch.tensor(d["ssim_loss"])
    x = img.clone().requires_grad_(True)
    s = ssim(x, gt)
    s.backward()
    out["ssim"], out["ssim_grad"] = s.detach(), x.grad.clone()
    out["ssim_map"] = ssim(img, gt, return_map=True)
    out["ssim_per_image"] = ssim(img, gt, size_average=False)
    # ---- regularization
    with cuda_as_cpu():
        V, E = 30, 140
        ei = torch.randint(0, V, (2, E), generator=g)
        rest = torch.randn(V, 3, generator=g)
        gauss = types.SimpleNamespace(mesh=types.SimpleNamespace(edge_index=ei),
                                      edge_norm=(rest[ei[1]] - rest[ei[0]]).norm(dim=-1, keepdim=True))
        ropt = types.SimpleNamespace(lambda_deform_mag=0.01, lambda_rigid=0.3, lambda_momentum=0.1)
        out.update(reg_edge_index=ei, reg_rest=rest)
        for T in (3, 2, 1):
            D = (rest[None] + 0.1 * torch.randn(T, V, 3, generator=g)).requires_grad_(True)
            loss = ns["regularization"](D, gauss, ropt)
            loss.backward()
            out[f"reg{T}.D"], out[f"reg{T}.loss"], out[f"reg{T}.grad"] = D.detach().clone(), loss.detach(), D.grad.clone()
        D = out["reg3.D"].clone().requires_grad_(True)
        out["reg3.static_loss"] = ns["regularization"](D, gauss, ropt, static=True).detach()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **{k: npy(v) for k, v in out.items()})


def gen_refine():
    """The `real_world` branch of the reference's rollout (train_meshnet_sim.py:212-250): the text of the branch's body is exec'd as it
    stands (ten iterations of a fresh Adam on the predicted velocities against the squared edge-length deviation, the
    `length_deviation[grasped_particle] *= 0` line included) on a small triangulated patch.  Stored: inputs and the refined velocities."""
    import torch.optim as optim
    lines = open(os.path.join(REF, "train_meshnet_sim.py")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip() == "if real_world:")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("predictions.append(predicted_next_velocity)"))
    body = "if True:\n" + "\n".join(lines[start + 1:end])        # (the branch's body as it stands, under a stand-in for `if real_world:`)
    g = torch.Generator().manual_seed(2125)
    out = {}
    for name, (gx, gy, amp) in {"a": (9, 7, 0.02), "b": (14, 11, 0.05)}.items():
        ys, xs = torch.meshgrid(torch.arange(gy, dtype=torch.float32), torch.arange(gx, dtype=torch.float32), indexing="ij")
        rest = torch.stack([xs.reshape(-1) * 0.05, ys.reshape(-1) * 0.05, torch.zeros(gx * gy)], 1)
        idx = lambda x, y: y * gx + x  # noqa: E731
        e = []
        for y in range(gy):
            for x in range(gx):
                for dx, dy in ((1, 0), (0, 1), (1, 1)):
                    if x + dx < gx and y + dy < gy:
                        e += [(idx(x, y), idx(x + dx, y + dy)), (idx(x + dx, y + dy), idx(x, y))]
        edge_index = torch.tensor(e, dtype=torch.long).t().contiguous()
        edge_index = edge_index[:, torch.randperm(edge_index.shape[1], generator=g)]
        original_edge_lengths = torch.norm(rest[edge_index.T[:, 1]] - rest[edge_index.T[:, 0]], dim=1)       # (:114-116)
        current_node_coords = rest + amp * torch.randn(rest.shape, generator=g)
        v = amp * torch.randn(rest.shape, generator=g)
        grasped = 5 if name == "a" else 37
        action = torch.tensor([[0.01, -0.02, 0.03]])
        ns = dict(torch=torch, optim=optim, predicted_next_velocity=v.clone(), current_node_coords=current_node_coords.clone(),
                  edge_index=edge_index, original_edge_lengths=original_edge_lengths, grasped_particle=grasped, current_action=action)
        exec(compile(body, "<train_meshnet_sim.real_world>", "exec"), ns)
        out.update({f"{name}.pos": current_node_coords, f"{name}.v": v, f"{name}.edge_index": edge_index, f"{name}.rest_len": original_edge_lengths,
                    f"{name}.grasped": torch.tensor(grasped), f"{name}.action": action, f"{name}.v_refined": ns["predicted_next_velocity"]})
    np.savez_compressed(os.path.join(OUT, "refine.npz"), **{k: npy(v) for k, v in out.items()})


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    only = sys.argv[1:]          # e.g. `python make_golden.py simulator`: regenerate the named fixtures only
    for name in ("camera", "sh", "misc", "normalizer", "gnn", "simulator", "densify", "scene_io", "meshsim", "mesh_transform", "losses",
                 "render_wiring", "gnn128", "vertice_rotation", "sim128", "refine"):
        if not only or name in only:
            globals()["gen_" + name]()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
