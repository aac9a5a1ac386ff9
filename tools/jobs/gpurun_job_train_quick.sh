# quick Row Z check: operator tests, cfg-4 bench, kernel trace of the step (no train_step tests)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PF_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_model.py -q -m gpu -k "train" 2>&1 | tail -3
timeout 600 python bench.py --config cfg4 --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/bench_cfg4.err | grep "^{" | tail -1 > gpurun_out/bench_cfg4.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg4.json').readline()); print('cfg4', round(d['value'],2), d['unit'], round(d['ms_per_step'],3), d.get('execution'))"
rm -rf gpurun_out/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 70 --title "cfg4 training step, steady state" | head -50 | cut -c1-150
rm -rf gpurun_out/prof_train
