for rep in 1 2; do
for d in . _old_r04; do
  (cd $d && python bench.py --eager --no-cpu-baseline --no-train-step --no-speculation $( [ "$d" = "." ] && echo --no-gnn ) --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['ms_per_step'], d['config']['launch'][:30], d.get('per_camera_ms_per_step'))")
done; done
