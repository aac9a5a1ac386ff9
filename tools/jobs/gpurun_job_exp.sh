cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_stages.py tests/test_gpu_teacher.py -m gpu -q -x -k "pyramid or test_mode or graphed_forward or stagewise or teacher or lanes or reference_model" --deselect "tests/test_gpu_stages.py::test_flow_iteration_stagewise_vs_oracle[cfg3-2]" --deselect "tests/test_gpu_teacher.py::test_teacher_forced_iterations_vs_oracle[cfg3-False]" > gpurun_out/pytest_x.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_x.log
timeout 300 python bench.py --no-cpu-baseline --calibration-steps 5 > gpurun_out/bench_x.log 2>&1
tail -5 gpurun_out/pytest_x.log; tail -1 gpurun_out/bench_x.log | cut -c1-330
