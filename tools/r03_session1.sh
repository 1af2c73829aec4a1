#!/bin/bash
# round-3 GPU session 1 (run ON THE GPU BOX from the repo root): new parity tests, the VALU-rate table, the N>1 bench lines on one
# GPU (gloo, two ranks sharing the device), the baseline bench line and the issue / traffic counters on the DEFAULT command.
set -u
TAG=${1:-r03a}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python3 -m pytest tests/test_raster_gpu.py -x -q -m gpu -k "config2_full_size" > "$OUT/test_config2.log" 2>&1; echo "config2 rc=$?" >> "$OUT/rc.txt"
timeout 600 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > "$OUT/test_dist_gpu.log" 2>&1; echo "dist_gpu rc=$?" >> "$OUT/rc.txt"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/valu_rate.hip -o /tmp/valu_rate && timeout 300 /tmp/valu_rate > "$OUT/valu_rate.txt" 2>&1; echo "valu_rate rc=$?" >> "$OUT/rc.txt"
timeout 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/rc.txt"
CSPLAT_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"; echo "bench2 rc=$?" >> "$OUT/rc.txt"
CSPLAT_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --mode scenes --steps 5 --warmup 2 > "$OUT/bench_2rank_scenes.json" 2> "$OUT/bench_2rank_scenes.err"; echo "scenes rc=$?" >> "$OUT/rc.txt"
sha1sum "$ROOT/cloth-splatting_amd/csrc/csplat_raster.hip" | cut -d" " -f1 > "$OUT/raster_src_sha1.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-train-step > "$OUT/trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-step > "$OUT/pmc_$c.log" 2>&1
done
cd "$ROOT" && bash tools/collect_issue_counters.sh "$TAG" > /dev/null 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
cat "$OUT/rc.txt"; tail -3 "$OUT/test_config2.log" "$OUT/test_dist_gpu.log"; cat "$OUT/valu_rate.txt"
