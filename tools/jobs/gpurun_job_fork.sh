cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
: > gpurun_out/fork_ab.log
for m in 0 1 2 0 1 2; do
  echo "== PF_FORK_MODE=$m" >> gpurun_out/fork_ab.log
  PF_FORK_MODE=$m timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step')})" >> gpurun_out/fork_ab.log
done
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
PF_FORK_MODE=$m timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/proftl -o g -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_tl.log 2>&1
python $R/tools/timeline.py $R/gpurun_out/proftl/g_results.db $R/gpurun_out/graph_timeline_mode$m.txt
rm -rf $R/gpurun_out/proftl
done
cat $R/gpurun_out/fork_ab.log
