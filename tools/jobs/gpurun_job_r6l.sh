# round-6 job l: the drop-in route with 2 / 3 / 4 / 6 worker processes on one GPU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 2 3 4 6; do
timeout 600 python bench.py --no-train-block --no-cpu-baseline --steps 5 --warmup 2 --calibration-steps 2 --route-workers $w > gpurun_out/bench_route_w$w.log 2>&1
grep "^{" gpurun_out/bench_route_w$w.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('workers $w', round(d['value'],1), json.dumps(d.get('route_reference_model'))[:900])"
done
