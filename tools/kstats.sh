#!/bin/bash
# tools/kstats.sh <n-rows> <python-script> [args...] -- rocprofv3 kernel trace of one python program (GPU box): per (kernel, grid)
# call count / total / average, top rows to stdout
N=$1; shift
export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstats.XXXX)
S=$(readlink -f "$1"); shift
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python3 "$S" "$@" > "$D/log" 2>&1
grep -v "^W2\|^E2\|amdgpu.ids" "$D/log" | tail -3
python3 - "$D" "$N" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:84], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2])]:
    print("%-84s wgs %6d calls %6d total %8.3f ms avg %8.1f us %5.2f%%" % (k[0], k[1], v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100.0 * v[1] / tot))
PY
