#!/usr/bin/env python3
"""kstats.py <kernel_stats.csv> <steps> [n]: per-step launch count / GPU time table of a rocprofv3 --stats run."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print(f"GPU ms/step {tot / 1e6 / steps:.3f}   launches/step {calls / steps:.1f}")
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e3 / steps:9.1f} us/step {float(r['AverageNs']) / 1e3:8.1f} us")
