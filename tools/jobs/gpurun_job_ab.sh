# same-box A/B of an environment switch: usage  AB_VAR=NAME AB_VALUES="a b" bash gpurun_job_ab.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab.log
for rep in 1 2; do
for v in $AB_VALUES; do
  echo "== $AB_VAR=$v (rep $rep)" >> gpurun_out/ab.log
  env PF_BENCH_GAP=1 $AB_VAR=$v timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step','host_issue_ms_per_step','gap_probe')})" >> gpurun_out/ab.log
done; done
cat gpurun_out/ab.log
