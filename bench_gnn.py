#!/usr/bin/env python3
"""bench_gnn.py -- BASELINE.json configs[3]: ClothMeshSimulator rollout, N = 10,000 nodes (100 x 100 cloth grid), E = 300,000
directed edges (every node receives from its 30 nearest neighbours in the plane -- a 2-D radius-like graph -- listed in PyG's
coalesced (row, col) order, i.e. NOT grouped by destination), latent 128, 15 message-passing steps, history 2 (nnode_in = 8),
eval mode: the rollout LOOP of /root/reference/train_meshnet_sim.py:126-265 (per step: edge features from the current
positions, predict_velocity, grasp pinning, integration; meshnet/rollout.py), 20 steps; plus the training step (forward +
backward + Adam).  Secondary bench (the driver's contract is bench.py); prints one JSON line.

Beside the HIP path it times a plain-torch restatement of what torch_geometric does on the GPU (index_select gathers,
[E,3L] concat, index_add_ scatter) with the same weights -- a same-device comparison of the data-movement design,
NOT the product path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def pyg_like_forward(net, x, ei, e):
    """what MessagePassing.propagate executes (gather, cat, MLP, scatter-add), for timing only"""
    x, e = net._encoder(x, e)
    for g in net._processor.gnn_stacks:
        xi, xj = x.index_select(0, ei[1]), x.index_select(0, ei[0])
        m = g.edge_fn(torch.cat([xi, xj, e], dim=-1))
        agg = torch.zeros_like(x).index_add_(0, ei[1], m)
        x = g.node_fn(torch.cat([agg, x], dim=-1)) + x
        e = e + e
    return net._decoder(x)


def cloth_graph(N, deg, gen):
    """nodes of a sqrt(N) x sqrt(N) cloth grid on [-0.5, 0.5]^2 (jittered by 20 % of the spacing, sinusoidal height as scene_1);
    every node receives an edge from each of its `deg` nearest neighbours in the plane; edge list in PyG's coalesced order
    (sorted by row = source, then col = target), as FaceToEdge / radius_graph hand it to the network"""
    import numpy as np
    from scipy.spatial import cKDTree
    g = int(round(N ** 0.5))
    assert g * g == N, "N must be a square"
    xs = torch.linspace(-0.5, 0.5, g)
    xy = torch.stack([xs.repeat(g), xs.repeat_interleave(g)], 1) + (torch.rand(N, 2, generator=gen) - 0.5) * (0.2 / (g - 1))
    pos = torch.cat([xy, 0.05 * torch.sin(3 * xy[:, :1]) * torch.cos(3 * xy[:, 1:])], 1).float()
    _, nb = cKDTree(xy.numpy()).query(xy.numpy(), k=deg + 1)
    src = torch.tensor(np.ascontiguousarray(nb[:, 1:]).reshape(-1))
    dst = torch.arange(N).repeat_interleave(deg)
    order = torch.argsort(src * N + dst)                         # coalesce: by (row, col)
    return pos, torch.stack([src[order], dst[order]])


def timeit(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=10_000)
    ap.add_argument("--deg", type=int, default=30)
    ap.add_argument("--no-train", action="store_true", help="rollout only (what bench.py embeds as its `gnn` field)")
    print(json.dumps(run(ap.parse_args())), flush=True)


def run(args, dev=None, train=None):
    """the measurement; returns the JSON record (bench.py embeds the rollout part as its `gnn` field)"""
    dev = dev or torch.device("cuda:0")
    train = (not getattr(args, "no_train", False)) if train is None else train
    from csplat import native
    from meshnet.cloth_network import ClothMeshSimulator
    torch.manual_seed(0)
    sim = ClothMeshSimulator(3, 8, 4, 128, 15, 2, 128, 2, 2, normalize=False, device=dev).eval()
    N, E = args.N, args.N * args.deg
    gen = torch.Generator().manual_seed(3)
    pos0, ei_cpu = cloth_graph(N, args.deg, gen)
    ei = ei_cpu.to(dev)
    pos0 = pos0.to(dev)
    vel = (torch.randn(N, 6, generator=gen) * 0.01).to(dev)
    ntype = torch.randint(0, 2, (N, 1), generator=gen).to(dev)
    from meshnet.rollout import edge_features, rollout
    ef = edge_features(pos0, ei)
    net = sim._encode_process_decode
    feats = torch.cat([vel, torch.nn.functional.one_hot(ntype.squeeze().long(), 2)], 1).float()

    nroll = 20
    actions = (torch.randn(nroll, 3, generator=gen) * 0.005).to(dev)
    hist = vel.reshape(N, 2, 3).permute(1, 0, 2).contiguous()

    def roll():
        return rollout(sim, pos0, hist, ntype, ei, actions, 0, nroll)
    from meshnet import rollout as ro
    from meshnet.graph_network import deferred_overflow_check
    with torch.no_grad():
        ms_loop = timeit(roll, max(args.steps // 10, 2), 1) / nroll          # ms per rollout step, whole loop (step 1 recorded, 18 replays)
        roll_stats = dict(ro.ROLLOUT_STATS)
        ms_loop_eager = timeit(lambda: rollout(sim, pos0, hist, ntype, ei, actions, 0, nroll, graph=False), 2, 1) / nroll
        # (predict_velocity reads its overflow word per call -- one host synchronisation; a loop of calls defers the reads, as rollout() does)
        with deferred_overflow_check() as chk:
            ms_roll = timeit(lambda: sim.predict_velocity(vel, ntype, ei, ef), args.steps, args.warmup)
            overflowed = chk.overflowed()
        ms_roll_sync = timeit(lambda: sim.predict_velocity(vel, ntype, ei, ef), args.steps, args.warmup)
        a = net(feats, ei, ef)
        ms_pyg = rel = None
        if train:
            ms_pyg = timeit(lambda: pyg_like_forward(net, feats, ei, ef), args.steps, args.warmup)
            b = pyg_like_forward(net, feats, ei, ef)
            rel = float((a - b).abs().max() / b.abs().max())
        native.prof_enable(["GNN"]); native.prof_read("GNN")
        sim.predict_velocity(vel, ntype, ei, ef); torch.cuda.synchronize()
        gnn_ms, gnn_n = native.prof_read("GNN"); native.prof_enable([])

    ms_train = ms_train_pyg = None
    if train:
        sim.train()
        opt = torch.optim.Adam(sim.parameters(), lr=1e-4)
        tgt = torch.randn(N, 3, device=dev) * 0.1

        def train_step(fwd):
            opt.zero_grad(set_to_none=True)
            pred = fwd()
            loss = ((pred - tgt) ** 2).mean()
            loss.backward()
            opt.step()
        ms_train = timeit(lambda: train_step(lambda: net(feats, ei, ef)), args.steps, args.warmup)
        ms_train_pyg = timeit(lambda: train_step(lambda: pyg_like_forward(net, feats, ei, ef)), args.steps, args.warmup)

    L, M = 128, 15
    alg = (16 * E + 12 * N) * L * M          # bytes of gather/scatter traffic per rollout step (the UNFUSED formulation's traffic)
    # what binds the fused layer kernels is the matrix pipe, not HBM: per layer the edge launch multiplies 3 Linears x
    # 2*128*128 x E in three fp16 piece-products, the node launch 6 matrices x 2*128*128 x N likewise
    piece_flop = M * 3 * 2 * 128 * 128 * (3 * E + 6 * N)
    MFMA_F16_DENSE = 2.5e15                  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
    # measured HBM bytes of the two layer launches (profiles/r05b_edge_mlp3_traffic.txt: FETCH_SIZE / WRITE_SIZE per edge launch at this
    # size, XCD-aware tile ranges: 182.0 MB read + 22.9 MB written; node launch: pieces in, three [N,128] out + x in = ~50 MB)
    hbm_bytes_step = M * (182.0e6 + 22.9e6 + 50.0e6) if (N, E) == (10_000, 300_000) else None
    r3 = lambda v: None if v is None else round(v, 3)  # noqa: E731
    out = {"metric": "MeshNet rollout step ms (N=10k, E=300k, L=128, M=15)", "value": round(ms_loop, 3), "unit": "ms",
           "higher_is_better": False, "dtype": "f32", "data": "synthetic",
           "rollout_loop_ms_per_step": round(ms_loop, 3), "rollout_loop_steps": nroll,
           "rollout_loop_eager_ms_per_step": round(ms_loop_eager, 3), "rollout_recording": roll_stats,
           "rollout_ms": round(ms_roll, 3), "rollout_ms_with_per_call_overflow_read": round(ms_roll_sync, 3), "overflowed": bool(overflowed),
           "pyg_like_torch_rollout_ms": r3(ms_pyg),
           "train_step_ms": r3(ms_train), "pyg_like_torch_train_step_ms": r3(ms_train_pyg),
           "hip_vs_pyg_like_rel_diff": rel,
           # algorithmic gather / scatter bytes of one rollout step over the step's time, against the 8 TB/s HBM peak
           "algorithmic_GBps": round(alg / (ms_loop * 1e-3) / 1e9, 1),
           # (NOT a utilisation: the unfused formulation's bytes over the fused kernels' time -- kept for continuity with rounds 1-5)
           "unfused_bytes_frac": round(alg / (ms_loop * 1e-3) / 1e9 / 8000.0, 4),
           "mfma_frac": round(piece_flop / (ms_loop * 1e-3) / MFMA_F16_DENSE, 4), "piece_product_flop_per_step": piece_flop,
           "hbm_GBps": None if hbm_bytes_step is None else round(hbm_bytes_step / (ms_loop * 1e-3) / 1e9, 1),
           "hbm_frac": None if hbm_bytes_step is None else round(hbm_bytes_step / (ms_loop * 1e-3) / 1e9 / 8000.0, 4),
           "hbm_bytes_per_step_from": "profiles/r05b_edge_mlp3_traffic.txt (PMC FETCH_SIZE / WRITE_SIZE of the edge launch) + the node launch's rows",
           "gnn_kernels": {"launches_per_step": int(gnn_n), "total_ms_per_step": round(gnn_ms, 3),
                           "algorithmic_GBps": round(alg / (gnn_ms * 1e-3) / 1e9, 1) if gnn_ms > 0 else None,
                           "algorithmic_bytes_per_step": alg},
           "config": {"workload": f"ClothMeshSimulator N={N} E={E} (2-D {args.deg}-nearest-neighbour graph on the cloth grid, coalesced "
                                  f"edge order) L=128 M=15 hist=2: {nroll}-step rollout loop (value), one predict_velocity call" +
                                  (", train step" if train else "")}}
    return out


if __name__ == "__main__":
    main()
