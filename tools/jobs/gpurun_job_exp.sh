cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/exp_cumask.py 2>&1 | grep -v Warning | tee gpurun_out/exp_cumask2.log | tail -16
