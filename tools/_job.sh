mkdir -p gpurun_out/r06
bash tools/ab_libs.sh 2 cur k8x > gpurun_out/r06/ab_k8_1.txt 2>&1; cat gpurun_out/r06/ab_k8_1.txt
python -m pytest tests/test_knn_gnn_gpu.py -q -p no:cacheprovider -k "rollout or refinement or overflow" > gpurun_out/r06/new_19.log 2>&1; tail -3 gpurun_out/r06/new_19.log
bash tools/gnn_kernel_table.sh r06_gnnprof2 2>&1 | grep -i "rollout_decode\|rows32\|absmax" | cut -c1-150
