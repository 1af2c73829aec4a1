mkdir -p gpurun_out/r06
python -m pytest tests/test_raster_gpu.py tests/test_config1.py -q -p no:cacheprovider -x > gpurun_out/r06/raster_10.log 2>&1; tail -5 gpurun_out/r06/raster_10.log
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-train-step --no-gnn --no-cpu-baseline --no-speculation --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_us'], d['roofline']['avg_launch_us'])"; done
