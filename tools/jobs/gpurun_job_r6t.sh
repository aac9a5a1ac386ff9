# round-6 job t: one-launch BatchNorm backward for planes that fit a block: tests, the cfg-4 step with / without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_model.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 900 -k "bn_relu or train_step or node or cfg4" > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for i in 1 2; do for v in PF_X=0 PF_BN_BWD_PLANE=0; do
env $v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 $v', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['batchnorm_backward']['kernel_us_per_step'], d['roofline']['batchnorm_backward']['launches_per_step'])"
done; done
