# round 5, first job: the cfg-4-size parity tests of the training step, the default bench line with its train block,
# the one-stream trace + ordered dispatch list of the training step
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_train gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train_cfg4.py tests/test_gpu_train_ops.py -m gpu -q --timeout 900 --durations=15 > gpurun_out/pytest_train.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_train.log
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 1200 --durations=10 -k "train_step or graphed or rmsprop" > gpurun_out/pytest_step.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_step.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_apply_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, steady state (hipGraph replay, PF_TRAIN_FORK=0: one stream)" > /dev/null
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
rm -rf gpurun_out/prof_train
tail -25 gpurun_out/pytest_train.log; tail -25 gpurun_out/pytest_step.log
python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg2.json').readline()); print(round(d['value'],2), d['unit'], json.dumps(d.get('train'))[:1500])"
