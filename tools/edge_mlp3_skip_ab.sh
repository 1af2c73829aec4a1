#!/bin/bash
# GPU box: where the one-launch edge MLP's time goes, by elimination.  ab/libcsplat_skip<mask>.so are builds with side tasks compiled out of
# the phases (bash csrc/build.sh -DEM_SKIP=<mask>: 1 ReLU + cut, 2 LayerNorm partials, 4 LayerNorm end + rows out, 8 gathers, 16 edge rows,
# 32 indices; results wrong); prints the kernel's time per variant, same box.   bash tools/edge_mlp3_skip_ab.sh 0 1 2 4 8 16 32 63
# The variant is selected through CSPLAT_LIB (csplat/native.py): the shipped libcsplat.so is never overwritten.
for m in "$@"; do
    echo -n "skip $m: "; CSPLAT_LIB=$PWD/ab/libcsplat_skip$m.so timeout 120 python3 tools/bench_edge_mlp3.py 2>/dev/null | grep "one launch" | tail -1
done
