#!/usr/bin/env python3
"""Same-process A/B of the config-4 rollout step with the edge MLP of every layer as ONE launch (csplat_gnn_edge_mlp3) and as the three
csplat_linear128 launches of rounds 1-4: alternated three times on one box."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench_gnn  # noqa: E402
import meshnet.graph_network as gn  # noqa: E402

for rep in range(3):
    for fused in (True, False):
        gn.EDGE_MLP_FUSED = fused
        r = bench_gnn.run(SimpleNamespace(steps=20, warmup=3, N=10_000, deg=30), torch.device("cuda:0"), train=False)
        print(f"rep {rep} fused={fused}: rollout {r['rollout_loop_ms_per_step']:.3f} ms/step, predict_velocity {r['rollout_ms']:.3f} ms, "
              f"{r['gnn_kernels']['launches_per_step']} launches", flush=True)
