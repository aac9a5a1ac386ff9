cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/exp_pipeline.log
for a in "--buffers 2" "--buffers 3" "--buffers 2 --split tower"; do
  echo "== $a" >> gpurun_out/exp_pipeline.log
  timeout 300 python -X faulthandler tools/exp_pipeline.py $a >> gpurun_out/exp_pipeline.log 2>&1; echo "exit $?" >> gpurun_out/exp_pipeline.log
done
grep -v amdgpu.ids gpurun_out/exp_pipeline.log | tail -40
