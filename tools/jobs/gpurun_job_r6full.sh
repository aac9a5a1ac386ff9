# round-6 full job (re-grounding at HEAD): whole GPU suite (no xfail marker left), smoke, default bench (headline printed
# before the train block's child), cfg-4 bench, rocprofv3 kernel trace of the 4-LANE GRAPH REPLAY folded per template
# instantiation, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) folded into
# pmc_traffic.json, the one-stream trace + ordered dispatch list of the cfg-4 training step, the backward microbench.
# Every step carries its own `timeout`.   SKIP_TESTS=1 / SKIP_PMC=1 / SKIP_TRAIN=1 shorten it.
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
echo "canary $(date) $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series')" > gpurun_out/canary.log
git -C $R rev-parse HEAD > gpurun_out/traced_sha.txt 2>/dev/null || echo "${PF_SHA:-unknown}" > gpurun_out/traced_sha.txt
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --durations=15 $TEST_ARGS > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
timeout 600 python bench.py --config cfg4 --no-cpu-baseline 2> gpurun_out/bench_cfg4.err | grep "^{" | tail -1 > gpurun_out/bench_cfg4.json
# kernel trace of the timed execution mode: 4 captured lanes, graph replay
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 4 --warmup 2 --calibration-steps 2 --no-cpu-baseline --no-train-block --no-extras > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/per_kernel_roofline.py summarize $DB gpurun_out/kernel_trace_cfg2_lanes4.json
find gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rocprof_kernel_stats_cfg2.csv \;
rm -rf gpurun_out/prof
if [ -z "$SKIP_PMC" ]; then
PMC_CMD="python $R/bench.py --eager --concurrency 0 --scenes-per-step 1 --steps 2 --warmup 1 --calibration-steps 1 --no-cpu-baseline --no-train-block"
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
if [ -z "$SKIP_TRAIN" ]; then
# cfg-4 step trace (one stream: with the flow tower on its second stream kernels overlap and their durations stop adding up)
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, steady state (hipGraph replay, PF_TRAIN_FORK=0: one stream)" > /dev/null
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
rm -rf gpurun_out/prof_train
timeout 600 python tools/microbench_train_ops.py > gpurun_out/microbench_train_ops.log 2>&1
fi
grep -E "passed|failed|error|XPASS|XFAIL" gpurun_out/pytest_gpu.log | tail -5; tail -2 gpurun_out/smoke.log
python -c "
import json
for f in ('gpurun_out/bench_cfg2.json','gpurun_out/bench_cfg4.json'):
    d=json.loads(open(f).readline()); print(f, round(d['value'],2), d['unit'], round(d['ms_per_step'],3), (d.get('cpu_baseline') or {}).get('kind'), json.dumps(d.get('roofline'))[:600])"
head -30 gpurun_out/per_kernel_roofline.md | cut -c1-200
