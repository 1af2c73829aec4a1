#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh r01
# Writes everything under gpurun_out/<tag>/ (merged back by gpurun); tools/summarize_profiles.py then turns it into the
# committed files under profiles/.  rocprofv3 gets the program itself after `--` and PMC passes carry no trace domains
# other than --kernel-trace, as the pool requires.
set -u
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 bench.py                     > "$OUT/bench.json"       2> "$OUT/bench.err"
# the N > 1 code paths of bench.py as far as ONE GPU allows: two ranks sharing the device, gloo as the transport (FlatGrads, the
# `collective` object with allreduce_ms / exposed_allreduce_ms) and the scene-parallel mode (6 scenes dealt over 2 ranks)
CSPLAT_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"
[ -n "${LIGHT:-}" ] || CSPLAT_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --mode scenes --steps 5 --warmup 2 > "$OUT/bench_2rank_scenes.json" 2> "$OUT/bench_2rank_scenes.err"
[ -n "${LIGHT:-}" ] || { /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/valu_rate.hip -o /tmp/valu_rate && timeout 300 /tmp/valu_rate > "$OUT/valu_rate.txt" 2>&1; }
python3 bench_gnn.py                 > "$OUT/bench_gnn.json"   2> /dev/null
python3 tools/gnn_train_trace.py 10  > "$OUT/gnn_train.txt"    2> /dev/null
python3 bench_train.py --steps 40 --warmup 5 > "$OUT/bench_train.json" 2> /dev/null
# the one-launch edge MLP against the three launches: kernel time, in-kernel phase stamps, same-box rollout A/B
{ python3 tools/bench_edge_mlp3.py; python3 tools/edge_mlp3_stamps.py; python3 tools/ab_edge_mlp3_rollout.py; } > "$OUT/edge_mlp3.txt" 2> /dev/null
# LIGHT=1: skip the micro-probes of kernels that did not change since the last full collection (GEMM / dW / rate probes, the scenes-mode 2-rank run)
if [ -z "${LIGHT:-}" ]; then
python3 tools/bench_linear128.py     > "$OUT/linear128.txt"    2> /dev/null
python3 tools/bench_linear128_train.py > "$OUT/linear128_train.txt" 2> /dev/null
python3 tools/bench_dw128.py 1000 10000 100000 300000 > "$OUT/dw128.txt" 2> /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate > "$OUT/mfma_rate.txt"
fi
cd /tmp && export TMPDIR=/tmp
# same command as the default bench (per-view streams on), so K7's average agrees with bench.py's HIP-event timing
# (--no-train-step: ONLY the headline workload's launches, so the per-kernel averages are one clean population)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-gnn --no-sustained > "$OUT/trace.log" 2>&1
# the same step with the views back to back on ONE stream: kernel durations free of cross-stream overlap (the profiler
# serialises dispatches of different streams more than a free run does, so only this pair of numbers can agree exactly)
python3 "$ROOT/bench.py" --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-view-streams > "$OUT/bench_serial.json" 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -o t -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-view-streams > "$OUT/trace_serial.log" 2>&1
# config 4: the product's training step alone (no PyG-like comparison leg in the population)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_gnn" -o t -- python3 "$ROOT/tools/gnn_train_trace.py" 10 > "$OUT/trace_gnn.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -o t -- python3 "$ROOT/bench_train.py" --steps 10 --warmup 3 > "$OUT/trace_train.log" 2>&1
sha1sum "$ROOT/cloth-splatting_amd/csrc/csplat_raster.hip" | cut -d" " -f1 > "$OUT/raster_src_sha1.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  # the DEFAULT command: the launches bench.py times (k_composite_bwd_views = all views of a step in one launch)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-speculation > "$OUT/pmc_$c.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmcserial_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-view-streams --no-speculation > "$OUT/pmcserial_$c.log" 2>&1
  [ -n "${LIGHT:-}" ] || timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_l128_$c" -o p -- python3 "$ROOT/tools/bench_linear128.py" 300000 2 > /dev/null 2>&1
done
cd "$ROOT" && bash tools/collect_issue_counters.sh "$TAG" > /dev/null 2>&1
# the summaries are formed HERE, on the box (gpurun merges at most 64 MiB back, and the raw per-dispatch counter files of one collection
# exceed that): gpurun_out/<tag>_profiles/ holds what tools/summarize_profiles.py would write into profiles/ -- copy it there
cd "$ROOT" && CSPLAT_PROFILES_DST="$ROOT/gpurun_out/${TAG}_profiles" python3 tools/summarize_profiles.py "$TAG" > "$OUT/summarize.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
du -sh "$OUT" "$ROOT/gpurun_out/${TAG}_profiles"
ls "$ROOT/gpurun_out/${TAG}_profiles"
