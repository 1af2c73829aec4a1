mkdir -p gpurun_out/r06
python -m pytest tests/test_raster_gpu.py -q -p no:cacheprovider -k "parts or sink" > gpurun_out/r06/new_6.log 2>&1; tail -15 gpurun_out/r06/new_6.log
python -m pytest tests/test_rccl_gpu.py -q -p no:cacheprovider > gpurun_out/r06/dist_6.log 2>&1; tail -10 gpurun_out/r06/dist_6.log
python bench.py --no-train-step --no-gnn --no-cpu-baseline > gpurun_out/r06/bench_6.json 2> gpurun_out/r06/bench_6.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/bench_6.json'))
for k in ("value","ms_per_step","eager_ms_per_step","per_camera_ms_per_step","sustained","speculation"):
    print(k, d[k])
PY
