#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of prebuilt library variants (ab/libcsplat_<name>.so), alternated ROUNDS times.
#   bash tools/ab_libs.sh ROUNDS name[:flags] name[:flags] ...      -> one line per run: name flags Mpix/s ms K7_us
# (figures from different boxes differ by +-4 %: only numbers from ONE call are comparable)
ROUNDS=$1; shift
LIB=cloth-splatting_amd/csplat/libcsplat.so
cp $LIB /tmp/libcsplat_keep.so
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    name=${v%%:*}; fl=0; [[ "$v" == *:* ]] && fl=${v##*:}
    cp ab/libcsplat_$name.so $LIB
    CSPLAT_DEBUG_FLAGS=$fl timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step --no-speculation 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', '$fl', d['value'], d['ms_per_step'], 'K7_us', d['roofline']['avg_launch_us'])"
  done
done
cp /tmp/libcsplat_keep.so $LIB
