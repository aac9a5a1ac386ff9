cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "edgeconv or edge" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_model.py -q -x -k "stage or lanes or golden" 2>&1 | tail -2
timeout 300 python tools/microbench_edge.py 2>/dev/null | tail -8
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --calibration-steps 4 --steps 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(round(d['value'],1), round(d['ms_per_depth_map'],4), 'edge_stats', round(k['pf_edge_stats_f32']['us_per_depth_map'],1), 'edge_apply', round(k['pf_edge_apply_f32']['us_per_depth_map'],1))"; done
