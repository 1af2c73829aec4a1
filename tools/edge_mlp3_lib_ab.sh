#!/bin/bash
# GPU box: same-box A/B of prebuilt library variants ab/libcsplat_<name>.so on tools/bench_edge_mlp3.py (kernel time of the one-launch edge MLP)
#   bash tools/edge_mlp3_lib_ab.sh ROUNDS name name ...
# The variant is selected through CSPLAT_LIB (csplat/native.py): the shipped libcsplat.so is never overwritten.
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
    for v in "$@"; do
        echo -n "$v: "; CSPLAT_LIB=$PWD/ab/libcsplat_$v.so timeout 120 python3 tools/bench_edge_mlp3.py 2>/dev/null | grep "one launch" | tail -1
    done
done
