#!/bin/bash
# GPU box: same-box A/B of prebuilt library variants ab/libcsplat_<name>.so on the config-4 rollout step (bench_gnn.py --no-train)
#   bash tools/rollout_lib_ab.sh ROUNDS name name ...
# The variant is selected through CSPLAT_LIB (csplat/native.py): the shipped libcsplat.so is never overwritten.
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
    for v in "$@"; do
        echo -n "$v: "; CSPLAT_LIB=$PWD/ab/libcsplat_$v.so timeout 200 python3 bench_gnn.py --no-train 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['rollout_loop_ms_per_step'], 'ms per rollout step;', d['gnn_kernels']['total_ms_per_step'], 'ms in GNN launches')"
    done
done
