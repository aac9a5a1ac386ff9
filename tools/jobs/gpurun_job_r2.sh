# round-2 GPU job: tests, smoke, bench, rocprof kernel trace, same-box A/B of the round's switches
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/parity_report.jsonl
if [ -z "$SKIP_TESTS" ]; then
eval "timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --timeout 600 $TEST_ARGS" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
timeout 600 python bench.py --steps 30 --warmup 3 $BENCH_ARGS > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
fi
if [ -n "$AB_LIST" ]; then
: > gpurun_out/ab.log
for rep in 1 2; do
for kv in $AB_LIST; do
  echo "== $kv (rep $rep)" >> gpurun_out/ab.log
  env $kv timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step')})
except Exception as e:
    print('failed', e)" >> gpurun_out/ab.log
done; done
fi
if [ -z "$SKIP_PROF" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
cd $R
python tools/rocprof_summary.py gpurun_out/prof/r1_results.db gpurun_out/kernel_stats.md --steps 9 --title "eager bench cfg2 (2 warm-up + 2 calibration + 5 timed steps)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline" --top 60
python tools/dispatch_list.py gpurun_out/prof/r1_results.db gpurun_out/last_step_dispatches.txt
rm -rf gpurun_out/prof
fi
if [ -n "$WITH_PMC" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_fetch.json
python tools/pmc_summary.py gpurun_out/pmc_write/w_results.db gpurun_out/pmc_write.json
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
fi
if [ -n "$EXTRA_CMD" ]; then
eval "$EXTRA_CMD" > gpurun_out/extra.log 2>&1
fi
cd $R
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-300; cat gpurun_out/ab.log 2>/dev/null
