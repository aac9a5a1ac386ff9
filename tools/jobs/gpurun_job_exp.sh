cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo "exit $?"; grep "^{" gpurun_out/bench_default.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['steps'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
