#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/trace_train.sh <tag>
# Kernel statistics of the config-3 train step (bench_train.py) -> gpurun_out/<tag>/trace_train/
set -u
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench_train.py > "$OUT/bench_train.json" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -o t -- python3 "$ROOT/bench_train.py" --steps 10 --warmup 3 > "$OUT/trace_train.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
cat "$OUT/bench_train.json"
