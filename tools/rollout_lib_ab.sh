#!/bin/bash
# GPU box: same-box A/B of prebuilt library variants ab/libcsplat_<name>.so on the config-4 rollout step (bench_gnn.py --no-train)
#   bash tools/rollout_lib_ab.sh ROUNDS name name ...
ROUNDS=$1; shift
LIB=cloth-splatting_amd/csplat/libcsplat.so
cp $LIB /tmp/libcsplat_keep.so
for r in $(seq $ROUNDS); do
    for v in "$@"; do
        cp ab/libcsplat_$v.so $LIB
        echo -n "$v: "; timeout 200 python3 bench_gnn.py --no-train 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['rollout_loop_ms_per_step'], 'ms per rollout step;', d['gnn_kernels']['total_ms_per_step'], 'ms in GNN launches')"
    done
done
cp /tmp/libcsplat_keep.so $LIB
