#!/usr/bin/env python3
"""tools/prof_train_host.py [pattern] -- cProfile of the config-3 train step's host side (GPU box): per-STEP own / cumulative
microseconds of every function of this repository (and torch entry points) that costs more than 3 us per step"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
import torch
import bench_train
from types import SimpleNamespace as NS
dev = torch.device("cuda:0")
bench_train.run(NS(steps=5, warmup=5, P=100_000, res=800, grid=100), dev)     # imports, caches
STEPS = 300
pr = cProfile.Profile()
pr.enable()
bench_train.run(NS(steps=STEPS, warmup=0, P=100_000, res=800, grid=100), dev)
pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if nc < STEPS // 2:
        continue
    rows.append((ct / STEPS * 1e6, tt / STEPS * 1e6, nc / STEPS, f"{os.path.basename(fn)}:{line}({name})"))
rows.sort(reverse=True)
print("  cum_us   own_us  calls/step  function")
for ct, tt, n, name in rows:
    if ct >= 3.0:
        print(f"{ct:8.1f} {tt:8.1f} {n:8.1f}   {name}")
