cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 30 --warmup 3 > gpurun_out/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/proftl -o g -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_tl.log 2>&1
cd $R
python tools/timeline.py gpurun_out/proftl/g_results.db gpurun_out/graph_timeline.txt
rm -rf gpurun_out/proftl
tail -1 gpurun_out/bench.log | cut -c1-250
