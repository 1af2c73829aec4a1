#!/bin/bash
# tools/k7_counters.sh <tag> [debug flags] -- PMC passes (issue counters) of the default bench command; prints K7's per-launch averages
set -u
TAG=${1:-k7c}; FL=${2:-0}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export CSPLAT_DEBUG_FLAGS=$FL
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
P3="SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/issue_$i" -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step > "$OUT/issue_$i.log" 2>&1
done
cd $ROOT
python3 tools/summarize_issue_counters.py $TAG 2>/dev/null | grep "batched k_composite_bwd" | cut -c1-2000
