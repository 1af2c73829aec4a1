#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_issue_counters.sh <tag>
# Instruction-issue counters of the rasterizer kernels: PMC-only passes (no trace domain besides
# --kernel-trace), <= 8 SQ counters per pass, of the DEFAULT bench command (one launch per stage for all views:
# k_composite_bwd_views / k_composite_fwd_views -> issue_*) and of the one-view-per-launch command (--no-view-streams ->
# issue_serial_*).  tools/summarize_issue_counters.py reduces them to profiles/<tag>_k67_issue.json.
set -u
TAG=${1:-r02}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
sha1sum "$ROOT/cloth-splatting_amd/csrc/csplat_raster.hip" | cut -d" " -f1 > "$OUT/raster_src_sha1.txt"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-gnn --no-sustained --no-speculation"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
P3="SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
P4="TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/issue_$i" -o p -- $CMD > "$OUT/issue_$i.log" 2>&1
done
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/issue_serial_$i" -o p -- $CMD --no-view-streams > "$OUT/issue_serial_$i.log" 2>&1
done
# SUMMARIZE=1: form the summary HERE, on the box, under gpurun_out/<tag>_profiles/ (copy it into profiles/), and drop the raw per-dispatch files
# (gpurun merges at most 64 MiB back)
if [ -n "${SUMMARIZE:-}" ]; then
  cd "$ROOT" && CSPLAT_PROFILES_DST="$ROOT/gpurun_out/${TAG}_profiles" python3 tools/summarize_issue_counters.py "$TAG" > "$OUT/summarize_issue.log" 2>&1
  find "$OUT" -name "*kernel_trace.csv" -delete
  find "$OUT" -name "*counter_collection.csv" -delete
  ls "$ROOT/gpurun_out/${TAG}_profiles"
else
  find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
  ls "$OUT"
fi
