cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "inverse or reproducible or edgeconv_autograd" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_backward_cfg4.py -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "train" 2>&1 | tail -4
timeout 300 python tools/microbench_edge_bwd.py 2>&1 | grep -v Warn | tee gpurun_out/microbench_edge_bwd.log
