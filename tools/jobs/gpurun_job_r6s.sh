# round-6 job s: whole GPU suite after the capture / garbage-collection guard and the latency-chain fixes (softargmin
# backward, flow-head weight sum, masked MAE, rows BatchNorm backward reduce); stand-alone numbers; the cfg-4 step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|error|exit" gpurun_out/pytest_gpu.log | tail -4
timeout 300 python tools/microbench_train_ops.py 2>&1 | grep -v "^wgrad\|amdgpu.ids" | tail -12
for i in 1 2; do
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['weight_gradients']['kernel_us_per_step'], d['roofline']['batchnorm_backward']['kernel_us_per_step'])"
done
