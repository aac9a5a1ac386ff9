# scene lanes: test + same-box A/B of --lanes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x --timeout 600 -k "lanes or graphed_forward or tiny" > gpurun_out/pytest_lanes.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_lanes.log
: > gpurun_out/lanes_ab.log; : > gpurun_out/lanes_err.log
for rep in 1 2; do for l in 1 2 3 4; do
  echo "== --lanes $l (rep $rep)" >> gpurun_out/lanes_ab.log
  timeout 300 python bench.py --lanes $l --no-cpu-baseline --calibration-steps 2 --steps 10 2>>gpurun_out/lanes_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_depth_map','host_issue_ms_per_depth_map','execution')})" >> gpurun_out/lanes_ab.log 2>&1
done; done
tail -5 gpurun_out/pytest_lanes.log; cat gpurun_out/lanes_ab.log
grep -v "amdgpu.ids" gpurun_out/lanes_err.log | tail -5
