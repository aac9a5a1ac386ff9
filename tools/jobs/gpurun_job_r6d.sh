# round-6 job d: the direct (vector-pipe) tower kernels: stand-alone per layer (matrix form / direct form), then the
# headline with PF_TOWER_DIRECT=0/1 interleaved on the same box, then the GPU tests that touch the towers with it on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/microbench_direct.log
for d in 0 1; do
echo "== PF_TOWER_DIRECT=$d" >> gpurun_out/microbench_direct.log
PF_TOWER_DIRECT=$d timeout 300 python tools/microbench_conv2d_wide.py 2>&1 | grep -E "conv0.1|conv1.0|conv1.1" >> gpurun_out/microbench_direct.log
done
cat gpurun_out/microbench_direct.log
if [ -z "$SKIP_AB" ]; then
AB_LIST="PF_TOWER_DIRECT=0 PF_TOWER_DIRECT=1" BENCH_ARGS="--no-train-block --no-extras" bash tools/jobs/gpurun_job_ab.sh
PF_TOWER_DIRECT=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --timeout 800 -k "tower or forward_test_mode or image_conv or conv2d" > gpurun_out/pytest_direct.log 2>&1; tail -4 gpurun_out/pytest_direct.log
fi
