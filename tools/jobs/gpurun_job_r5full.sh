# round-5 job for the FIRST run with GPU access again: a canary first (a box that cannot even echo is not this repo's
# fault), then the whole GPU suite (the xfail-marked file last: XPASS / XFAIL per test with -rxX), smoke, the default
# bench (headline + train block + experiments from its child process), the two experiments' microbench, the 4-lane kernel
# trace folded per instantiation, and the one-stream trace + ordered dispatch list of the training step.
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/prof_train gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
echo "canary $(date) $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series')" > gpurun_out/canary.log
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=15 $TEST_ARGS > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
fi
timeout 1200 python bench.py --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
timeout 600 python tools/microbench_split.py > gpurun_out/microbench_split.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 4 --warmup 2 --calibration-steps 2 --no-cpu-baseline --no-train-block > $R/gpurun_out/rocprof.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/per_kernel_roofline.py summarize $DB gpurun_out/kernel_trace_cfg2_lanes4.json
python tools/per_kernel_roofline.py report gpurun_out/kernel_trace_cfg2_lanes4.json gpurun_out/per_kernel_roofline --config cfg2 > /dev/null
rm -rf gpurun_out/prof
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_apply_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, steady state (hipGraph replay, PF_TRAIN_FORK=0: one stream)" > /dev/null
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
rm -rf gpurun_out/prof_train
grep -E "passed|failed|error|XPASS|XFAIL" gpurun_out/pytest_gpu.log | tail -40; tail -2 gpurun_out/smoke.log
python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg2.json').readline()); print(round(d['value'],2), d['unit'], json.dumps(d.get('train'))[:3000])"
cat gpurun_out/microbench_split.log
