#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of prebuilt library variants (ab/libcsplat_<name>.so), alternated ROUNDS times.
#   bash tools/ab_libs.sh ROUNDS name[:flags] name[:flags] ...      -> one line per run: name flags Mpix/s ms K7_us
# (figures from different boxes differ by +-4 %: only numbers from ONE call are comparable)
# The variant is selected through CSPLAT_LIB (csplat/native.py): the shipped libcsplat.so is never touched.
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    name=${v%%:*}; fl=0; [[ "$v" == *:* ]] && fl=${v##*:}
    CSPLAT_BENCH_ELIMINATION_BUILD=1 CSPLAT_LIB=$PWD/ab/libcsplat_$name.so CSPLAT_DEBUG_FLAGS=$fl timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step --no-gnn --no-speculation --no-sustained 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', '$fl', d['value'], d['ms_per_step'], 'K7_us', d['roofline']['avg_launch_us'], 'K8_us', d['kernel_us']['K8_preprocess_bwd'])"
  done
done
