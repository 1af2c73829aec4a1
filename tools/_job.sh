mkdir -p gpurun_out/r06
python -m pytest tests/test_train_gpu.py -q -p no:cacheprovider -k "captured" > gpurun_out/r06/captured_1.log 2>&1; tail -30 gpurun_out/r06/captured_1.log
python -m pytest tests/test_raster_gpu.py -q -p no:cacheprovider -k "sink or reproducible or bit_repro or 225" > gpurun_out/r06/sink_1.log 2>&1; tail -15 gpurun_out/r06/sink_1.log
python tools/calibrate_captured_atomic.py --pairs 36 > gpurun_out/r06/calib.log 2>&1; tail -3 gpurun_out/r06/calib.log
cp tests/golden/captured_atomic_calibration.json profiles/r06_captured_atomic_calibration.txt gpurun_out/r06/
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/r06/full_1.log 2>&1; tail -40 gpurun_out/r06/full_1.log
python bench.py > gpurun_out/r06/bench_1.json 2> gpurun_out/r06/bench_1.err; cat gpurun_out/r06/bench_1.json
