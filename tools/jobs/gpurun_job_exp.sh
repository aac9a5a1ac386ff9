cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap | tee gpurun_out/mfma_valu_overlap.log
