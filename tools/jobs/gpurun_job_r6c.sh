# round-6 job c: the drop-in route after the module graphs: its tests, its bench line (fast on / off), its profile
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 800 -k "module_graphs or reference_model_py or library_convolution or drop_in or hip_route" > gpurun_out/pytest_route.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_route.log
REF=oracle/_ref/reference_model_py.txt
for fast in 1 0 1; do
PF_DROPIN_FAST=$fast timeout 300 python bench.py --route reference-model --reference-model-py $REF --no-cpu-baseline --no-extras --steps 10 --warmup 3 2> gpurun_out/route_$fast.err | grep "^{" | tail -1 > gpurun_out/bench_route_fast$fast.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_route_fast$fast.json').readline()); print('PF_DROPIN_FAST=$fast', round(d['value'],1), d['unit'], round(d['ms_per_depth_map'],2))"
done
timeout 600 python tools/profile_route.py --maps 12 --out gpurun_out/route_profile_fast.md > gpurun_out/route_profile_fast.log 2>&1
tail -8 gpurun_out/pytest_route.log; head -3 gpurun_out/route_profile_fast.md; tail -3 gpurun_out/route_1.err
