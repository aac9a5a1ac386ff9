# round-6 job y: EdgeConv backward in two walks (PF_EDGE_BWD_SUMS) + the one-launch look-back scan of the counting sorts:
# operator / step tests, same-box A/B of the cfg-4 step, the one-stream trace of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 600 -k "edgeconv or knn_inverse or gather_knn or sort" > gpurun_out/pytest_ops.log 2>&1; tail -3 gpurun_out/pytest_ops.log
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward_cfg4.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 900 > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for i in 1 2 3; do for v in 1 0; do
PF_EDGE_BWD_SUMS=$v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 edge_sums $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
rm -rf gpurun_out/prof_train
PF_TRAIN_FORK=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, steady state (one stream)" | head -12 | cut -c1-150
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel" > /dev/null
grep -E 'edge_bwd|scan_chain' gpurun_out/cfg4_last_step_dispatches.txt | awk '{print $1, $3, $11, $12, $13}' | cut -c1-110
rm -rf gpurun_out/prof_train
