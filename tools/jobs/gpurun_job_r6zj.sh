# round-6 job zj: scene lanes in flight after the XCD-aware block order (4 was best before it)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for lanes in 4 5 6 3 8; do
timeout 300 python bench.py --steps 10 --warmup 3 --lanes $lanes --no-cpu-baseline --no-train-block --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg2 lanes $lanes', round(d['value'],1), round(d['ms_per_step'],3))"
done; done 2>&1 | tee gpurun_out/lanes_after_xcd.log
