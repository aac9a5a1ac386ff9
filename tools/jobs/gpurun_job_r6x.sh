# round-6 job x: the PointFlow nodes' 1x1 weight gradients deferred to the END of the backward (PF_WGRAD_LATE): tests, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_model.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 900 -k "train_step or node or weight_gradient or wgrad or cfg4" > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for i in 1 2 3; do for v in 1 0; do
PF_WGRAD_LATE=$v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 late $v', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['weight_gradients']['kernel_us_per_step'], d['roofline']['weight_gradients']['launches_per_step'])"
done; done
