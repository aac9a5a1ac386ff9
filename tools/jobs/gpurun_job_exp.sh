cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out


for lanes in 4 1; do for c in 0 2; do [ $lanes = 4 -a $c = 0 ] && continue
echo "== lanes=$lanes PF_CONCURRENCY=$c"
PF_CONCURRENCY=$c timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 --lanes $lanes 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4), 'towers', d['roofline']['towers'], 'frac', d['roofline']['frac'])"
done; done
