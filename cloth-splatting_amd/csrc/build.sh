#!/bin/bash
# Builds libcsplat.so (gfx950 only) next to this script's parent: cloth-splatting_amd/csplat/libcsplat.so
# One object per .hip file under csrc/build/ (recompiled when the source, a header or this script is newer), compiled in
# parallel, then linked.  FORCE=1 rebuilds everything.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../csplat/libcsplat.so"
OBJ="$HERE/build"
mkdir -p "$OBJ"
# -fno-slp-vectorize: v_pk_{mul,add,fma}_f32 cost a SIMD what two plain instructions cost (profiles/r04b_valu_rate.txt), but forming their
# register pairs costs moves -- and copies of just-loaded registers that drag the load's s_waitcnt to the front of a loop (K7, round 4).
# For every file: same-box A/B of the GNN kernels with and without it -- plain / gather `k_linear128` unchanged (73-74 / 97-99 us), the
# LayerNorm variant 93.5 -> 88.5 us, rollout 5.37 -> 5.33 ms (figures from different boxes differ by more than that: 75 vs 80 us).
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
pids=()
objs=()
for src in "$HERE"/*.hip; do
    o="$OBJ/$(basename "${src%.hip}").o"
    objs+=("$o")
    stale=0
    if [ -n "$FORCE" ] || [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$0" -nt "$o" ]; then stale=1; fi
    for h in "$HERE"/*.h "$HERE"/../../include/*.h; do [ "$h" -nt "$o" ] && stale=1; done
    if [ $stale = 1 ]; then
        /opt/rocm/bin/hipcc $FLAGS "$@" -c "$src" -o "$o" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $OUT"
