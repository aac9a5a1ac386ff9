# round-6 job q: swapped-operand weight gradients (conv0_1, conv6_2) + coalesced split reduce: tests, stand-alone, step A/B;
# the partial-store share under the new plans (PF_WGRAD_DBG=4); the leak fixture (route test, then the frustum test)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --timeout 800 -k "module_graphs or frustum_variance_vs_reference" > gpurun_out/pytest_leak.log 2>&1; tail -2 gpurun_out/pytest_leak.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_ops.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 800 -k "train_step or node or weight_gradient or wgrad or cfg4" > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for v in PF_X=0 PF_WGRAD_SWAP=0 PF_WGRAD_DBG=4; do
echo "== $v"; env $v timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad\|^# weight" > gpurun_out/mb_$v.log; tail -1 gpurun_out/mb_$v.log
done
paste <(cut -c1-52 gpurun_out/mb_PF_X=0.log) <(cut -c34-46 gpurun_out/mb_PF_WGRAD_SWAP=0.log) <(cut -c34-46 gpurun_out/mb_PF_WGRAD_DBG=4.log) | head -20
for i in 1 2; do for v in PF_X=0 PF_WGRAD_SWAP=0; do
env $v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 $v', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['weight_gradients']['kernel_us_per_step'])"
done; done
