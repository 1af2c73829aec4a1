mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r06/full_12.log 2>&1; tail -6 gpurun_out/r06/full_12.log
python bench.py > gpurun_out/r06/bench_12.json 2> gpurun_out/r06/bench_12.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/bench_12.json'))
for k in ("value","ms_per_step","eager_ms_per_step","per_camera_ms_per_step","kernel_us","sustained","gnn","train_step"):
    print(k, d.get(k))
print(d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["atomics"])
PY
