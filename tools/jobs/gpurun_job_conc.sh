cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for c in 0 1 2 3; do
  PF_CONCURRENCY=$c timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conc $c rep $rep', round(d['value'],1), round(d['ms_per_step'],3))"
done
done | tee gpurun_out/concurrency.log
