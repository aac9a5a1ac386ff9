# round-6 job r: channel-fastest partial layout: tests, stand-alone table, the cfg-4 step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_ops.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 900 -k "train_step or node or weight_gradient or wgrad or cfg4" > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad\|^# weight" > gpurun_out/microbench_wgrad.log; cat gpurun_out/microbench_wgrad.log | cut -c1-60
for i in 1 2; do
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['weight_gradients']['kernel_us_per_step'])"
done
