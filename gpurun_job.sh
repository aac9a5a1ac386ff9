cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 30 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
cd $R
tail -8 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-1500
