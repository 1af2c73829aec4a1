#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/trace_bench.sh <tag>  -> gpurun_out/<tag>/trace_bench/ (kernel trace of bench.py)
set -u
TAG=${1:-tb}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_bench" -o t -- python3 "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-train-step > "$OUT/trace_bench.log" 2>&1
ls -la "$OUT/trace_bench"/* | head
