LIB=cloth-splatting_amd/csplat/libcsplat.so
cp $LIB /tmp/keep.so
cp ab/libcsplat_p2.so $LIB
TEACHER_DUMP=gpurun_out/teacher_fail.npz timeout 100 python -m pytest tests/test_psnr_parity_gpu.py -m gpu -x -q -s -k "reproducible_k7" 2>&1 | grep -E "^step|passed|failed" | cut -c1-200
cp /tmp/keep.so $LIB
