mkdir -p gpurun_out/r06
python -m pytest tests/test_knn_gnn_gpu.py tests/test_raster_gpu.py -q -p no:cacheprovider -k "refinement or overflow or sink or rollout or encoder" > gpurun_out/r06/new_4.log 2>&1; tail -30 gpurun_out/r06/new_4.log
python bench_gnn.py --no-train > gpurun_out/r06/bench_gnn_4.json 2> gpurun_out/r06/bench_gnn_4.err; tail -3 gpurun_out/r06/bench_gnn_4.json; tail -5 gpurun_out/r06/bench_gnn_4.err
