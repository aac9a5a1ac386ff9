# round-3 job B: re-run of the reworked parity tests + the scene-concurrency experiment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_backward_cfg4.py "tests/test_gpu_stages.py" \
   "tests/test_gpu_model.py::test_forward_test_mode_vs_reference" "tests/test_gpu_model.py::test_cfg2_full_size_properties" \
   -m gpu -q --timeout 900 --durations=8 > gpurun_out/pytest_r3b.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_r3b.log
timeout 600 python tools/exp_two_graphs.py > gpurun_out/exp_two_graphs.log 2>&1; echo "exit $?" >> gpurun_out/exp_two_graphs.log
tail -12 gpurun_out/pytest_r3b.log; cat gpurun_out/exp_two_graphs.log | tail -15
