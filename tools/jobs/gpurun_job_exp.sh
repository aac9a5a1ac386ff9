cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=pointmvsnet_amd/build/variants
for arm in base td3 td4 td2w2; do cp $V/lib_$arm.so pointmvsnet_amd/libpointflow_hip.so; echo "== $arm: $(timeout 200 python tools/microbench_conv3d_pair.py 2>/dev/null | tr '\n' ' ')"; done
cp $V/lib_base.so pointmvsnet_amd/libpointflow_hip.so
LIB_LIST='base td3 td4' LANES_LIST=4 bash tools/jobs/gpurun_job_libab.sh | tail -12
