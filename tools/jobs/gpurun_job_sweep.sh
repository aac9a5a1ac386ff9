cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/sweep.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "lanes or graphed_forward" 2>&1 | tail -2 >> gpurun_out/sweep.log
for rep in 1 2; do for l in 1 2 3 4 8; do
  echo -n "lanes=$l : " >> gpurun_out/sweep.log
  PF_WIDE_GLOBALB=1 timeout 300 python bench.py --lanes $l --no-cpu-baseline --calibration-steps 2 --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4), d['lane_placement_probe_maps_per_s'])" >> gpurun_out/sweep.log 2>&1
done; done
cat gpurun_out/sweep.log
