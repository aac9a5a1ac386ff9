# round-6 job za: XCD-aware tile order of the EdgeConv gather passes (PF_EDGE_XCD): tests, same-box A/B of the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py -m gpu -q -x --timeout 600 -k "edge or stage or flow" > gpurun_out/pytest_ops.log 2>&1; tail -3 gpurun_out/pytest_ops.log
for i in 1 2 3; do for v in 1 0; do
PF_EDGE_XCD=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-block --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('cfg2 edge_xcd $v', round(d['value'],1), round(d['ms_per_step'],3), 'stats', round(k['pf_edge_stats_f32']['us_per_depth_map'],1), 'apply', round(k['pf_edge_apply_f32']['us_per_depth_map'],1))"
done; done 2>&1 | tee gpurun_out/edge_xcd_ab.log
