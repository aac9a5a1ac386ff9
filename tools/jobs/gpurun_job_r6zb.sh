# round-6 job zb: XCD-aware block order per kernel family (PF_XCD mask builds, tools/experiments/build_xcd_variants.sh):
# parity tests on the default build (all families on), then same-box A/B of the headline per mask
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py tests/test_gpu_model.py -m gpu -q -x --timeout 900 -k "not reference_model_py" > gpurun_out/pytest_ops.log 2>&1; tail -3 gpurun_out/pytest_ops.log
for rep in 1 2; do for m in 0 1 2 4 255; do
PF_LIB_PATH=$GRAFT_REPO_ROOT/tools/experiments/libpointflow_XCD$m.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-block --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('cfg2 xcd mask $m', round(d['value'],1), round(d['ms_per_step'],3), ' '.join('%s %.1f' % (n[3:-4], k[n]['us_per_depth_map']) for n in ('pf_conv2d_wide_sets_f32','pf_conv3d_k3_pair_f32','pf_conv3d_k3_f32','pf_flow_features_f32') if n in k))"
done; done 2>&1 | tee gpurun_out/xcd_mask_ab.log
