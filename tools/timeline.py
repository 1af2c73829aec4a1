#!/usr/bin/env python3
"""timeline.py <kernel_trace.csv> [marker=k_adam]: kernels of one steady-state step (between marker launches) with start
offset, duration, queue -- to see where the GPU idles waiting for the host."""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_adam'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-5] + 1, idx[-3] + 1
t0 = int(rows[a]['Start_Timestamp'])


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('at::native::', '')
    m = re.search(r'(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|reduce_kernel|index_elementwise_kernel|'
                  r'unrolled_elementwise_kernel)<[^>]*?([A-Za-z_]+(Functor|Ops|kernel_cuda|_kernel|functor)[A-Za-z_<>]*)', n)
    return (m.group(1)[:12] + ':' + m.group(2)[:40]) if m else n[:60]


busy, end = 0, 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    busy += max(0, e - max(s, end)); end = max(end, e)
    print(f"{s / 1e3:8.1f} {(e - s) / 1e3:6.1f} q{r['Queue_Id']} {r['Grid_Size_X']:>9} {short(r['Kernel_Name'])}")
print(f"span {end / 1e3:.1f} us, busy {busy / 1e3:.1f} us, launches {b - a}")
