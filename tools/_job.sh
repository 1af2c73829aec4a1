mkdir -p gpurun_out/r06
python -m pytest tests/test_knn_gnn_gpu.py tests/test_reference_goldens_gpu.py -q -p no:cacheprovider > gpurun_out/r06/new_16.log 2>&1; tail -30 gpurun_out/r06/new_16.log
python bench_gnn.py --no-train > gpurun_out/r06/bench_gnn_16.json 2> gpurun_out/r06/bench_gnn_16.err; tail -2 gpurun_out/r06/bench_gnn_16.json | cut -c1-700
