cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/wide_ab.log
for v in 0 1; do
  echo "== PF_WIDE_GLOBALB=$v microbench" >> gpurun_out/wide_ab.log
  PF_WIDE_GLOBALB=$v timeout 300 python tools/microbench_conv2d_wide.py 2>&1 | grep "views" | cut -c1-100 >> gpurun_out/wide_ab.log
done
cat gpurun_out/wide_ab.log
