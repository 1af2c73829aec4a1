#!/bin/bash
# Builds libcsplat.so (gfx950 only) next to this script's parent: cloth-splatting_amd/csplat/libcsplat.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../csplat/libcsplat.so"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -o "$OUT" "$HERE"/csplat_sort.hip "$HERE"/csplat_raster.hip "$HERE"/csplat_knn.hip "$HERE"/csplat_gnn.hip "$HERE"/csplat_image.hip "$HERE"/csplat_mesh.hip "$HERE"/csplat_gemm.hip "$HERE"/csplat_optim.hip "$HERE"/csplat_sim.hip "$@"
echo "built $OUT"
