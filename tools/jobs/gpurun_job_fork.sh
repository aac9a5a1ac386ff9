# same-box A/B of the training step on one stream / flow tower forked / weight gradients forked too, + the whole-step tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PF_MIOPEN_FIND=0
mkdir -p gpurun_out; : > gpurun_out/fork_ab.log
for rep in 1 2; do for heads in 1; do for fork in "2 0" "2 3"; do set -- $fork
  echo "== fork=$fork batch=$heads rep $rep" >> gpurun_out/fork_ab.log
  PF_TRAIN_FORK=$1 PF_WGRAD_FORK=$2 timeout 600 python bench.py --config cfg4 --no-cpu-baseline --steps 20 --warmup 3 2> gpurun_out/bench_cfg4.err | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), d['unit'], round(d['ms_per_step'],3), d.get('execution'))" >> gpurun_out/fork_ab.log 2>&1
done; done; done
cat gpurun_out/fork_ab.log; tail -5 gpurun_out/bench_cfg4.err
grep train_step_gradients gpurun_out/parity_report.jsonl | tail -2
