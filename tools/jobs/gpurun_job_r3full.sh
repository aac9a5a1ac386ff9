# round-3 full job: whole GPU suite, smoke, bench (graph, lanes), rocprof kernel stats of the eager forward
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/parity_report.jsonl
if [ -z "$SKIP_TESTS" ]; then
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=12 $TEST_ARGS > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
fi
timeout 600 python bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --eager --concurrency 0 --scenes-per-step 1 --steps 5 --warmup 2 --calibration-steps 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
cd $R
python tools/rocprof_summary.py gpurun_out/prof/r1_results.db gpurun_out/kernel_stats.md --steps 11 --title "eager bench cfg2 (2 warm-up + 2 calibration + 2 + 5 forwards)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --eager --concurrency 0 --scenes-per-step 1 --steps 5 --warmup 2 --calibration-steps 2 --no-cpu-baseline" --top 60
python tools/dispatch_list.py gpurun_out/prof/r1_results.db gpurun_out/last_step_dispatches.txt
rm -rf gpurun_out/prof
if [ -n "$WITH_PMC" ]; then
cd /tmp
PMC_CMD="python $R/bench.py --eager --concurrency 0 --scenes-per-step 1 --steps 2 --warmup 1 --calibration-steps 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f -- $PMC_CMD > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w -- $PMC_CMD > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_fetch.json
python tools/pmc_summary.py gpurun_out/pmc_write/w_results.db gpurun_out/pmc_write.json
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
fi
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-300; head -30 gpurun_out/kernel_stats.md
