mkdir -p gpurun_out/r06
python -m pytest tests/test_raster_gpu.py tests/test_config1.py -q -p no:cacheprovider -x > gpurun_out/r06/raster_7.log 2>&1; tail -15 gpurun_out/r06/raster_7.log
python bench.py --no-train-step --no-gnn --no-cpu-baseline > gpurun_out/r06/bench_7.json 2> gpurun_out/r06/bench_7.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/bench_7.json'))
for k in ("value","ms_per_step","eager_ms_per_step","per_camera_ms_per_step","kernel_us","replay_check"):
    print(k, d[k])
print(d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
PY
