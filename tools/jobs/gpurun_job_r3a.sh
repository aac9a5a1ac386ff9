# round-3 job A: the new parity tests (teacher-forced, cfg5 / cfg4 goldens, cfg4-size backward, cfg3 stage case), smoke, bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_backward_cfg4.py tests/test_eval_output.py tests/test_bench_launch.py \
   "tests/test_gpu_model.py::test_forward_test_mode_vs_reference" "tests/test_gpu_model.py::test_forward_train_mode_no_grad_vs_reference" \
   "tests/test_gpu_model.py::test_forward_autograd_path_vs_reference_and_backward" "tests/test_gpu_model.py::test_graphed_forward_matches_eager_and_replays_on_new_scenes" \
   tests/test_gpu_stages.py -m gpu -q --timeout 900 --durations=15 > gpurun_out/pytest_r3a.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_r3a.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -25 gpurun_out/pytest_r3a.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-600
