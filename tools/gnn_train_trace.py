#!/usr/bin/env python3
"""tools/gnn_train_trace.py [steps] -- ONLY the product's config-4 train step (forward + backward + Adam), for
`rocprofv3 --kernel-trace --stats -- python3 tools/gnn_train_trace.py`: a clean per-kernel population of the training path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench_gnn  # noqa: E402
from meshnet.cloth_network import ClothMeshSimulator  # noqa: E402
from meshnet.rollout import edge_features  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
sim = ClothMeshSimulator(3, 8, 4, 128, 15, 2, 128, 2, 2, normalize=False, device=dev).train()
gen = torch.Generator().manual_seed(3)
N = 10_000
pos0, ei = bench_gnn.cloth_graph(N, 30, gen)
ei, pos0 = ei.to(dev), pos0.to(dev)
ef = edge_features(pos0, ei)
feats = torch.cat([torch.randn(N, 6, generator=gen) * 0.01, torch.nn.functional.one_hot(torch.randint(0, 2, (N,), generator=gen), 2)], 1).float().to(dev)
net = sim._encode_process_decode
opt = torch.optim.Adam(sim.parameters(), lr=1e-4)
tgt = torch.randn(N, 3, device=dev) * 0.1


def step():
    opt.zero_grad(set_to_none=True)
    loss = ((net(feats, ei, ef) - tgt) ** 2).mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"train step {(time.perf_counter() - t0) / steps * 1e3:.3f} ms")
