# round-6 job b: the drop-in route's profile (host timeline + cProfile + torch profiler; rocprofv3 kernel stats), the teacher-forced
# parity tests with step (C) at cfg 3 / 4 / 5, and the default bench line with its new first-class keys
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_route gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
timeout 600 python tools/profile_route.py --maps 12 --out gpurun_out/route_profile.md > gpurun_out/route_profile.log 2>&1
PROFILE=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_route -o rr -- python $R/tools/profile_route.py --maps 12 > $R/gpurun_out/route_rocprof.log 2>&1
find gpurun_out/prof_route -name "*kernel_stats.csv" -exec cp {} gpurun_out/route_kernel_stats.csv \;
rm -rf gpurun_out/prof_route
timeout 1500 python -m pytest tests/test_gpu_teacher.py -m gpu -q --timeout 1400 > gpurun_out/pytest_teacher.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_teacher.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
tail -5 gpurun_out/pytest_teacher.log; head -5 gpurun_out/route_profile.md; head -25 gpurun_out/route_kernel_stats.csv | cut -c1-160
python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg2.json').readline())
print({k:d.get(k) for k in ('value','value_pcie_inclusive','value_one_lane','extras_note','unique_scenes_cycled','route_reference_model','cpu_baseline_cfg1')})
print((d.get('train') or {}).get('value'), (d.get('train') or {}).get('ms_per_step'), (d.get('train') or {}).get('dispatches_per_step'), json.dumps((d.get('train') or {}).get('experiments'))[:600])"
