# round-4 full job: whole GPU suite, smoke, default bench, kernel trace of the 4-LANE GRAPH REPLAY (per-instantiation roofline),
# PMC traffic per instantiation, cfg-4 bench + trace, microbench of the backward kernels
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --durations=12 $TEST_ARGS > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
fi
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
timeout 600 python bench.py --config cfg4 --no-cpu-baseline 2> gpurun_out/bench_cfg4.err | grep "^{" | tail -1 > gpurun_out/bench_cfg4.json
# kernel trace of the timed execution mode: 4 captured lanes, graph replay
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 4 --warmup 2 --calibration-steps 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/per_kernel_roofline.py summarize $DB gpurun_out/kernel_trace_cfg2_lanes4.json
rm -rf gpurun_out/prof
if [ -n "$WITH_PMC" ]; then
PMC_CMD="python $R/bench.py --eager --concurrency 0 --scenes-per-step 1 --steps 2 --warmup 1 --calibration-steps 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f -- $PMC_CMD > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w -- $PMC_CMD > $R/gpurun_out/pmc_write.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/pmc_fetch -name "*.db" | head -1) gpurun_out/pmc_fetch.json
python tools/pmc_summary.py $(find gpurun_out/pmc_write -name "*.db" | head -1) gpurun_out/pmc_write.json
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
python tools/per_kernel_roofline.py report gpurun_out/kernel_trace_cfg2_lanes4.json gpurun_out/per_kernel_roofline --config cfg2 --pmc-fetch gpurun_out/pmc_fetch.json --pmc-write gpurun_out/pmc_write.json > /dev/null
else
python tools/per_kernel_roofline.py report gpurun_out/kernel_trace_cfg2_lanes4.json gpurun_out/per_kernel_roofline --config cfg2 > /dev/null
fi
# cfg-4 step trace
# (one stream: with the flow tower on its second stream kernels overlap and their durations stop adding up)
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_apply_kernel<64" --per-step 2 --steps 2 --top 80 --title "cfg4 training step, steady state (hipGraph replay, PF_TRAIN_FORK=0: one stream)" > /dev/null
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
rm -rf gpurun_out/prof_train
timeout 600 python tools/microbench_train_ops.py > gpurun_out/microbench_train_ops.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; tail -2 gpurun_out/smoke.log
python -c "
import json
for f in ('gpurun_out/bench_cfg2.json','gpurun_out/bench_cfg4.json'):
    d=json.loads(open(f).readline()); print(f, round(d['value'],2), d['unit'], round(d['ms_per_step'],3), (d.get('cpu_baseline') or {}).get('kind'), json.dumps(d.get('roofline'))[:600])"
head -30 gpurun_out/per_kernel_roofline.md | cut -c1-200
