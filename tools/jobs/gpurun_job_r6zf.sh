# round-6 job zf: the inverted-list gather with chunked ids / 8 pairs in flight: tests, cfg-4 step, one-stream trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py tests/test_gpu_backward_cfg4.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 900 > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for i in 1 2 3; do
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4', round(d['value'],1), round(d['ms_per_step'],3))"
done 2>&1 | tee gpurun_out/cfg4_runs.log
rm -rf gpurun_out/prof_train
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, steady state (one stream)" | head -8 | cut -c1-150
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
grep -E 'resize_bwd' gpurun_out/cfg4_last_step_dispatches.txt | awk '{print $1, $3, $11, $12, $13}' | cut -c1-110
rm -rf gpurun_out/prof_train
