# phase ablation of the tower kernels: LIB_LIST names library variants under pointmvsnet_amd/build/variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/phases.log
V=pointmvsnet_amd/build/variants
for arm in $LIB_LIST; do
  cp $V/lib_$arm.so pointmvsnet_amd/libpointflow_hip.so
  echo "== $arm: $(timeout 300 python tools/microbench_wide_phases.py 2>/dev/null | tail -1)" >> gpurun_out/phases.log
done
cp $V/lib_base.so pointmvsnet_amd/libpointflow_hip.so
cat gpurun_out/phases.log
