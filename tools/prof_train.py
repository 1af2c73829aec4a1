#!/usr/bin/env python3
"""Host-side profile (cProfile) of the timed train steps of bench_train.py: python tools/prof_train.py"""
import cProfile
import io
import os
import pstats
import sys

sys.argv = ["bench_train.py", "--steps", "40", "--warmup", "8"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_train  # noqa: E402
from csplat import train as tr  # noqa: E402

pr = cProfile.Profile()
orig = tr.train_step
calls = [0]


def wrapped(*a, **k):
    calls[0] += 1
    if calls[0] <= 8:
        return orig(*a, **k)
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()


tr.train_step = wrapped
bench_train.main()
for key, n in (("tottime", 40), ("cumulative", 70)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue())
