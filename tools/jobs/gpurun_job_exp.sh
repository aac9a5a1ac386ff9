cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "tower or image_conv or conv2d" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "lanes or graphed_forward" 2>&1 | tail -3
for lanes in 4 4; do
echo "== lanes=$lanes"
timeout 300 python bench.py --no-cpu-baseline --calibration-steps 4 --steps 10 --lanes $lanes 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4), 'towers', d['roofline']['towers'], 'frac', d['roofline']['frac']); print({k:(v['launches_per_depth_map'], round(v['us_per_depth_map'],1)) for k,v in d['kernels'].items() if 'conv2d' in k})"
done
