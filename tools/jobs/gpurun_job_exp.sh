cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/exp_two_graphs.log
for a in "--lanes 2" "--lanes 3" "--lanes 4" "--onegraph 2" "--onegraph 3"; do
  echo "== $a" >> gpurun_out/exp_two_graphs.log
  timeout 300 python tools/exp_two_graphs.py $a >> gpurun_out/exp_two_graphs.log 2>&1; echo "exit $?" >> gpurun_out/exp_two_graphs.log
done
grep -v amdgpu.ids gpurun_out/exp_two_graphs.log
