cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/exp_queues.log
for e in "X=1" "EXP_IMPORT_BENCH=1" "EXP_SET_DEVICE=1" "EXP_LOAD_LIB=1" "PF_CONCURRENCY=2"; do
  echo "== $e" >> gpurun_out/exp_queues.log
  env PF_WIDE_GLOBALB=1 PF_CONCURRENCY=0 $e timeout 400 python tools/exp_queues.py >> gpurun_out/exp_queues.log 2>&1
done
grep -v amdgpu.ids gpurun_out/exp_queues.log
