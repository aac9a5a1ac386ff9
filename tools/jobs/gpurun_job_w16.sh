cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "conv2d_wide" 2>&1 | tail -2
for tpb in 1 2 3 4; do
echo "== PF_WIDE16_TPB=$tpb"
PF_WIDE16_TPB=$tpb timeout 300 python tools/microbench_conv2d_wide.py 2>&1 | grep "^3 views conv[01]"
done
