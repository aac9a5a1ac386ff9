cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "inverse or reproducible or edgeconv_autograd or gather_knn" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_backward_cfg4.py tests/test_gpu_model.py -q -x -k "cfg4 or train" 2>&1 | tail -3
timeout 300 python tools/microbench_edge_bwd.py 2>&1 | grep -v Warn | tee gpurun_out/microbench_edge_bwd.log | tail -8
PF_MIOPEN_FIND=0 timeout 400 python bench.py --config cfg4 --no-cpu-baseline --steps 5 --warmup 3 > gpurun_out/bench_cfg4_nofind.log 2>&1; grep "^{" gpurun_out/bench_cfg4_nofind.log | tail -1 > gpurun_out/bench_cfg4_nofind.json; tail -2 gpurun_out/bench_cfg4_nofind.log | cut -c1-400
timeout 400 python bench.py --route reference-model --reference-model-py oracle/_ref/reference_model_py.txt --no-cpu-baseline > gpurun_out/bench_route.log 2>&1; grep "^{" gpurun_out/bench_route.log | tail -1 > gpurun_out/bench_route_reference_model.json; tail -1 gpurun_out/bench_route.log | cut -c1-300
