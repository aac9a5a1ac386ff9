cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gather_knn or edgeconv_autograd or inverse" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "train or autograd" 2>&1 | tail -4
