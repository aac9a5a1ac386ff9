# Row Z job: the training step's own kernels -- operator tests, train-step tests, cfg-4 bench, kernel trace of the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PF_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_ops.py -q -m gpu 2>&1 | tail -40 > gpurun_out/train_ops_test.log
tail -15 gpurun_out/train_ops_test.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "train_step" 2>&1 | tail -40 > gpurun_out/train_step_test.log
tail -12 gpurun_out/train_step_test.log
timeout 600 python bench.py --config cfg4 --no-cpu-baseline --steps 5 --warmup 3 2> gpurun_out/bench_cfg4.err | grep "^{" | tail -1 > gpurun_out/bench_cfg4.json
tail -3 gpurun_out/bench_cfg4.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg4.json').readline()); print('cfg4', round(d['value'],2), d['unit'], round(d['ms_per_step'],3), d.get('execution'))"
rm -rf gpurun_out/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 60 --title "cfg4 training step, steady state" | head -90
rm -rf gpurun_out/prof_train
