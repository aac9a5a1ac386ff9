# round-6 job zm: the late weight gradients on 1-4 streams (PF_WGRAD_STREAMS): gradient tests with 3, same-box sweep of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
PF_WGRAD_STREAMS=3 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_train_cfg4.py -m gpu -q -x --timeout 600 -k "train_step" 2>&1 | tail -2
for rep in 1 2 3; do for v in 1 2 3 4; do
PF_WGRAD_STREAMS=$v timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 wgrad_streams $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done 2>&1 | tee gpurun_out/wgrad_streams_ab.log
