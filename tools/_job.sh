mkdir -p gpurun_out/r06
for i in 1 2 3; do
  python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06/final_$i.log 2>&1
  tail -3 gpurun_out/r06/final_$i.log
done
