cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/fv.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "frustum or fetch" 2>&1 | tail -3 >> gpurun_out/fv.log
timeout 300 python tools/microbench_frustum.py 2>&1 | grep -v amdgpu | cut -c1-100 >> gpurun_out/fv.log
cat gpurun_out/fv.log
