#!/bin/bash
# GPU box: hardware counters of k_edge_mlp3r under tools/bench_edge_mlp3.py, one rocprofv3 --pmc pass per group (no trace domains beside
# --kernel-trace); prints the per-dispatch mean of every counter for that kernel and leaves nothing but the table under gpurun_out/.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/em_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
G3="TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_READ_TAGCONFLICT_STALL_CYCLES"
G4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"
i=0
for g in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/bench_edge_mlp3.py" > "$OUT/p$i.log" 2>&1
    f="$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)"
    [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$OUT/table.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "edge_mlp3r" not in r.get("Kernel_Name", ""):
        continue
    a = acc[r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"{k:36s} mean per dispatch {s / n:16.1f}   ({n} dispatches)")
PY
    rm -rf "$OUT/p$i"
done
