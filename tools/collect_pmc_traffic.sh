#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_pmc_traffic.sh <tag>
# Only the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh (HBM bytes per launch of every rasterizer kernel) and their summary.
set -u
TAG=${1:-r05a}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
sha1sum "$ROOT/cloth-splatting_amd/csrc/csplat_raster.hip" | cut -d" " -f1 > "$OUT/raster_src_sha1.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-speculation > "$OUT/pmc_$c.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmcserial_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-view-streams --no-speculation > "$OUT/pmcserial_$c.log" 2>&1
done
cd "$ROOT" && CSPLAT_PROFILES_DST="$ROOT/gpurun_out/${TAG}_profiles" python3 tools/summarize_profiles.py "$TAG" > "$OUT/summarize_pmc.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
ls "$ROOT/gpurun_out/${TAG}_profiles"
