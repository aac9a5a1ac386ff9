# same-box A/B of environment switches: usage  AB_LIST="A=1 B=2,C=3 ..." (comma joins several variables of one arm)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab.log
for rep in 1 2; do
for arm in $AB_LIST; do
  echo "== $arm (rep $rep)" >> gpurun_out/ab.log
  env $(echo $arm | tr ',' ' ') timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4))" >> gpurun_out/ab.log 2>&1
done; done
cat gpurun_out/ab.log
