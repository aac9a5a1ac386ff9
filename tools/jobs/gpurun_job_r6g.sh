# round-6 job g: the shape-gated matrix-core ConvTranspose3d's tests; the weight-gradient kernel taken apart
# (PF_WGRAD_DBG: 1 = no staging, 2 = no MFMA loop, 4 = no partial store) at config 4's shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 500 -k "deconv3d" > gpurun_out/pytest_deconv.log 2>&1; tail -3 gpurun_out/pytest_deconv.log
: > gpurun_out/wgrad_ablation.log
for v in 0 1 2 4 3 6 7; do
echo "== PF_WGRAD_DBG=$v" >> gpurun_out/wgrad_ablation.log
PF_WGRAD_DBG=$v timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad" >> gpurun_out/wgrad_ablation.log
done
python - <<'P'
import re, collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/wgrad_ablation.log"):
    if l.startswith("=="):
        cur = l.split("=")[-1].strip(); continue
    m = re.match(r"wgrad (.*?)\s+([\d.]+) us", l)
    if m: rows.setdefault(m.group(1).strip(), {})[cur] = float(m.group(2))
print("%-28s %s" % ("layer", "  ".join("dbg%s" % k for k in "0124367")))
for k, v in rows.items():
    print("%-28s %s" % (k, "  ".join("%5.1f" % v.get(c, -1) for c in "0124367")))
P
