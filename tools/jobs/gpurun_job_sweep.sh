cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/sweep.log
for cfg in cfg1 cfg3 cfg5; do for l in 1 2 4; do
  echo -n "$cfg lanes=$l : " >> gpurun_out/sweep.log
  timeout 600 python bench.py --config $cfg --lanes $l --no-cpu-baseline --calibration-steps 2 --steps 6 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['ms_per_depth_map'],4))" >> gpurun_out/sweep.log 2>&1
done; done
cat gpurun_out/sweep.log
