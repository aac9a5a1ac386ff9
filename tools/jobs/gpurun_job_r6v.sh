# round-6 job v: PointFlow weight gradients (EdgeConv chain, MLP) issued on the side stream (PF_TRAIN_FORK=3) against 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
PF_TRAIN_FORK=3 timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x --timeout 800 -k "train_step" > gpurun_out/pytest_fork3.log 2>&1; tail -2 gpurun_out/pytest_fork3.log
for i in 1 2 3; do for v in 2 3; do
PF_TRAIN_FORK=$v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 fork $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
