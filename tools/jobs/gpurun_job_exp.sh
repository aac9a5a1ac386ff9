cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "independent_scenes or lanes" 2>&1 | tail -12
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "tower" 2>&1 | tail -3
for cfgs in "4 1" "4 2" "2 2" "2 4" "4 3" "3 2"; do set -- $cfgs
echo "== lanes=$1 scenes-per-call=$2"
timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 --scenes-per-step 48 --lanes $1 --scenes-per-call $2 2>gpurun_out/err.log | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4))" || tail -3 gpurun_out/err.log
done
