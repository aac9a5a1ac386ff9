# end-of-round job: gpurun_job_r3full.sh (suite, smoke, default bench, kernel stats, PMC traffic) + the other configs,
# the stage timeline of one lane and the drop-in route
cd $GRAFT_REPO_ROOT
WITH_PMC=1 bash tools/jobs/gpurun_job_r3full.sh
cd $GRAFT_REPO_ROOT
for cfg in cfg1 cfg3 cfg5; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_$cfg.json
done
PF_MIOPEN_FIND=0 timeout 400 python bench.py --config cfg4 --no-cpu-baseline --steps 5 --warmup 3 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_cfg4_nofind.json
PF_TIMELINE=1 timeout 300 python bench.py --lanes 1 --no-cpu-baseline --steps 20 > gpurun_out/bench_timeline.log 2>&1
timeout 300 python bench.py --lanes 1 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_cfg2_lanes1.json
timeout 400 python bench.py --route reference-model --reference-model-py oracle/_ref/reference_model_py.txt --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_route_reference_model.json
for f in gpurun_out/bench_cfg*.json gpurun_out/bench_route_reference_model.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); print('$f', round(d['value'],2), d['unit'], round(d['ms_per_step'],3))"; done
grep "^{" gpurun_out/bench_timeline.log | tail -1 > gpurun_out/bench_timeline.json; python -c "import json; d=json.loads(open('gpurun_out/bench_timeline.json').readline()); print(d['stage_timeline_us'])"
