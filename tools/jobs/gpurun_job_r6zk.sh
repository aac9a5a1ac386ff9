# round-6 job zk: band order for the coarse warp (PF_XCD bit 32) against the default build: parity test + same-box A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
PF_LIB_PATH=$GRAFT_REPO_ROOT/tools/experiments/libpointflow_XCD33.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x --timeout 600 -k "frustum or forward_test_mode or variance" 2>&1 | tail -2
for rep in 1 2 3; do for m in default 33; do
if [ $m = default ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$GRAFT_REPO_ROOT/tools/experiments/libpointflow_XCD$m.so; fi
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-block --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('cfg2 xcd $m', round(d['value'],1), round(d['ms_per_step'],3), 'frustum', round(k['pf_frustum_variance_cl_f32']['us_per_depth_map'],1))"
done; done 2>&1 | tee gpurun_out/xcd_frustum_ab.log
