cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "bottom" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/dispatch_list.py gpurun_out/prof/r1_results.db gpurun_out/last_step_dispatches.txt
rm -rf gpurun_out/prof
grep -i "bottom" gpurun_out/last_step_dispatches.txt | cut -c1-120
