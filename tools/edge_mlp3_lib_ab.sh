#!/bin/bash
# GPU box: same-box A/B of prebuilt library variants ab/libcsplat_<name>.so on tools/bench_edge_mlp3.py (kernel time of the one-launch edge MLP)
#   bash tools/edge_mlp3_lib_ab.sh ROUNDS name name ...
ROUNDS=$1; shift
LIB=cloth-splatting_amd/csplat/libcsplat.so
cp $LIB /tmp/libcsplat_keep.so
for r in $(seq $ROUNDS); do
    for v in "$@"; do
        cp ab/libcsplat_$v.so $LIB
        echo -n "$v: "; timeout 120 python3 tools/bench_edge_mlp3.py 2>/dev/null | grep "one launch" | tail -1
    done
done
cp /tmp/libcsplat_keep.so $LIB
