# round-6 job h: producer / consumer weight-gradient kernel: its tests, the stand-alone table (+ ablation), the cfg-4 step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x --timeout 800 -k "weight_gradient" > gpurun_out/pytest_wgrad.log 2>&1; tail -3 gpurun_out/pytest_wgrad.log
: > gpurun_out/wgrad_ablation.log
for v in 0 1 2 4; do
echo "== PF_WGRAD_DBG=$v" >> gpurun_out/wgrad_ablation.log
PF_WGRAD_DBG=$v timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad\|^# weight" >> gpurun_out/wgrad_ablation.log
done
python - <<'P'
import re, collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/wgrad_ablation.log"):
    if l.startswith("=="):
        cur = l.split("=")[-1].strip(); continue
    m = re.match(r"wgrad (.*?)\s+([\d.]+) us", l)
    if m: rows.setdefault(m.group(1).strip(), {})[cur] = float(m.group(2))
    if l.startswith("#"): print(cur, l.strip())
print("%-28s %s" % ("layer", "  ".join("dbg%s" % k for k in "0124")))
for k, v in rows.items():
    print("%-28s %s" % (k, "  ".join("%5.1f" % v.get(c, -1) for c in "0124")))
P
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_ops.py -m gpu -q -x --timeout 800 -k "train_step or node or weight_gradient" > gpurun_out/pytest_train.log 2>&1; tail -3 gpurun_out/pytest_train.log
for i in 1 2; do
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['weight_gradients']['kernel_us_per_step'])"
done
