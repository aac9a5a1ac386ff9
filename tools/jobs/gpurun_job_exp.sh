cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/mb_edge.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py -m gpu -q -k "edgeconv or stagewise" --deselect "tests/test_gpu_stages.py::test_flow_iteration_stagewise_vs_oracle[cfg3-2]" 2>&1 | tail -12 >> gpurun_out/mb_edge.log
for v in 0 1; do echo "== PF_EDGE_LAT=$v" >> gpurun_out/mb_edge.log; PF_EDGE_LAT=$v timeout 300 python tools/microbench_edge.py 2>&1 | grep -v amdgpu >> gpurun_out/mb_edge.log; done
cat gpurun_out/mb_edge.log
