#!/usr/bin/env python3
"""tools/prof_train_host.py -- cProfile of the config-3 train step's host side (GPU box): top functions by own time, per step"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd")); sys.path.insert(0, ROOT)
sys.argv = ["bench_train.py", "--steps", "300", "--warmup", "20"]
import bench_train
pr = cProfile.Profile()
pr.enable()
bench_train.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
out = s.getvalue()
print(out[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:12000])
