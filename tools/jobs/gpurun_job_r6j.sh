# round-6 job j: weight-gradient plan experiments (4x4x16 tiles for 3-D layers; fewer rows per block = fewer position slices)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/wgrad_plans.log
for v in PF_X=0 PF_WGRAD_TILE3D=44 PF_WGRAD_MT_CAP=1 PF_WGRAD_MT_CAP=2; do
echo "== $v" >> gpurun_out/wgrad_plans.log
env $v timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad\|^# weight" >> gpurun_out/wgrad_plans.log
done
python - <<'P'
import re, collections
rows = collections.OrderedDict(); cur = None; cols=[]
for l in open("gpurun_out/wgrad_plans.log"):
    if l.startswith("=="):
        cur = l[3:].strip(); cols.append(cur); continue
    m = re.match(r"wgrad (.*?)\s+([\d.]+) us", l)
    if m: rows.setdefault(m.group(1).strip(), {})[cur] = float(m.group(2))
    if l.startswith("# weight gradients per"): print(cur, l.strip())
print("%-28s %s" % ("layer", "  ".join("%18s" % k for k in cols)))
for k, v in rows.items():
    print("%-28s %s" % (k, "  ".join("%18.1f" % v.get(c, -1) for c in cols)))
P
