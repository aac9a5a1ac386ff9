# round-6 job e: quick validation of the Python-side changes (module-graph staleness checks, bench keys): the route's tests +
# bench line, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 800 -k "module_graphs or reference_model_py or library_convolution" > gpurun_out/pytest_route.log 2>&1; tail -3 gpurun_out/pytest_route.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg2.json
python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg2.json').readline())
print({k:d.get(k) for k in ('value','value_pcie_inclusive','value_one_lane','extras_note','parity')})
print(d.get('route_reference_model'))
print((d.get('train') or {}).get('value'), (d.get('train') or {}).get('ms_per_step'), (d.get('train') or {}).get('dispatches_per_step'))"
