# end-of-round job: tests, smoke, bench (graph), stage timeline, rocprof kernel stats, PMC traffic, other configs
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
# PMC first (its JSON feeds roofline.traffic of the bench line)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_fetch.json
python tools/pmc_summary.py gpurun_out/pmc_write/w_results.db gpurun_out/pmc_write.json
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json gpurun_out/pmc_traffic.json
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 python bench.py --steps 30 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
PF_TIMELINE=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_timeline.json
for cfg in cfg1 cfg3 cfg5; do
timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$cfg.json
done
# (cfg4 is not part of this job: its MIOpen find step costs ~6 GPU-minutes; see profiles/r02al_cfg4_train_bench.txt)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof/r1_results.db gpurun_out/kernel_stats.md --steps 9 --title "eager bench cfg2 (2 warm-up + 2 calibration + 5 timed steps)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline" --top 60
python tools/dispatch_list.py gpurun_out/prof/r1_results.db gpurun_out/last_step_dispatches.txt
rm -rf gpurun_out/prof
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-200
for f in gpurun_out/bench_cfg*.json gpurun_out/bench_timeline.json; do echo $f; cut -c1-160 $f; done
