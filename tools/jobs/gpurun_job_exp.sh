cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "tower or image_conv or conv2d" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "lanes or golden or reference or graphed_forward" 2>&1 | tail -2
LIB_LIST='base tpb4new' LANES_LIST='1' bash tools/jobs/gpurun_job_libab.sh | tail -8
cp pointmvsnet_amd/build/variants/lib_tpb4new.so pointmvsnet_amd/libpointflow_hip.so
