cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "hip_route or reference_model_py or batch_of_two or train_step_runs" > gpurun_out/pytest_route.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_route.log
timeout 600 python bench.py --route reference-model --reference-model-py oracle/_ref/reference_model_py.txt --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_route_reference_model.json
tail -8 gpurun_out/pytest_route.log; cut -c1-700 gpurun_out/bench_route_reference_model.json
