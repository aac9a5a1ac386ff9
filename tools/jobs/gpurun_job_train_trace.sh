# cfg-4 step on ONE stream: ordered dispatch list of the last step + folded statistics (attribution of the small launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PF_MIOPEN_FIND=0 PF_TRAIN_FORK=${PF_TRAIN_FORK:-0}
mkdir -p gpurun_out; rm -rf gpurun_out/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o r4 -- python bench.py --config cfg4 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python tools/last_steps_stats.py $DB gpurun_out/cfg4_last_steps_1stream.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2 --top 90 --title "cfg4 training step, one stream" > /dev/null
python tools/dispatch_list.py $DB gpurun_out/cfg4_last_step_dispatches.txt "conv3d_k3_pair_kernel"
rm -rf gpurun_out/prof_train
grep -c . gpurun_out/cfg4_last_step_dispatches.txt
