# round-6 job f: the matrix-core ConvTranspose3d (deconv3d_k3s2_mfma_kernel): stand-alone against the lane-per-cell form
# (PF_DECONV_VALU=1), its tests, then the headline and the training step with it on / off on the same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/microbench_deconv.log
for v in "" 1; do
echo "== PF_DECONV_VALU=$v" >> gpurun_out/microbench_deconv.log
env ${v:+PF_DECONV_VALU=$v} timeout 300 python tools/microbench_deconv3d.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/microbench_deconv.log
done
cat gpurun_out/microbench_deconv.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_train_ops.py -m gpu -q -x --timeout 800 -k "deconv or volume or forward_test_mode or train_step" > gpurun_out/pytest_deconv.log 2>&1; tail -3 gpurun_out/pytest_deconv.log
AB_LIST="PF_DECONV_VALU=1 PF_X=0" BENCH_ARGS="--no-train-block --no-extras" bash tools/jobs/gpurun_job_ab.sh
for v in 1 ""; do
env ${v:+PF_DECONV_VALU=$v} timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 PF_DECONV_VALU=$v', round(d['value'],1), round(d['ms_per_step'],3))"
done
