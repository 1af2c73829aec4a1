#!/bin/bash
# GPU box: HBM traffic of the one-launch edge MLP (k_edge_mlp3r) per launch, from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate
# passes, no trace domain beside --kernel-trace), under tools/bench_edge_mlp3.py with EM_AGG=1 (what the rollout runs) and without.
# gfx950 correction as for the rasterizer (MI355X_MICROARCH.md): FETCH_SIZE tallies 64 B per 128-B request -> doubled; KB units.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/em_traffic"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for agg in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EM_AGG=$agg timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/p" -o p -- python3 "$ROOT/tools/bench_edge_mlp3.py" > "$OUT/log_${agg}_$c.txt" 2>&1
    f="$(find "$OUT/p" -name '*counter_collection.csv' | head -1)"
    [ -n "$f" ] && python3 - "$f" "$agg" "$c" <<'PY' | tee -a "$OUT/table.txt"
import csv, sys
tot = n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "edge_mlp3r" in r.get("Kernel_Name", "") and r["Counter_Name"] == sys.argv[3]:
        tot += float(r["Counter_Value"]); n += 1
kb = tot / max(n, 1)
mb = kb * (2.0 if sys.argv[3] == "FETCH_SIZE" else 1.0) * 1024.0 / 1e6
print(f"aggregation {'on ' if sys.argv[2] == '1' else 'off'} {sys.argv[3]:10s} {kb:12.1f} KB per launch -> {mb:8.1f} MB ({n} launches)")
PY
    rm -rf "$OUT/p"
  done
done
