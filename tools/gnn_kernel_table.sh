#!/bin/bash
# Per-kernel times of the config-4 rollout (bench_gnn.py --no-train) under rocprofv3 --kernel-trace; prints the top of the stats table and
# leaves only the summary under gpurun_out/ (the raw trace is deleted on the box: gpurun_out/ merges back only up to 64 MiB).
# GPU box: bash tools/gnn_kernel_table.sh [name]
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="${1:-gnnprof}"
OUT="$ROOT/gpurun_out/$NAME"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/bench_gnn.py" --no-train --steps 20 > "$OUT/run.log" 2>&1
f="$(find "$OUT" -name '*kernel_stats.csv' | head -1)"
if [ -n "$f" ]; then
    cp "$f" "$OUT/kernel_stats.csv"
    python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:88]:88s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
else
    echo "no kernel_stats.csv produced"; tail -5 "$OUT/run.log"
fi
find "$OUT" -type f ! -name kernel_stats.csv ! -name run.log -delete
tail -1 "$OUT/run.log" | cut -c1-600
