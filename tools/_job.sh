mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06/full_13.log 2>&1; tail -3 gpurun_out/r06/full_13.log
LIGHT=1 bash tools/collect_profiles.sh r06a > gpurun_out/r06/collect_r06a.log 2>&1; tail -30 gpurun_out/r06/collect_r06a.log
