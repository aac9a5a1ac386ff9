# round-6 job i: start skew of the weight-gradient kernel's resident rounds (PF_WGRAD_SKEW shader cycles per round)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/wgrad_skew.log
for v in 0 2000 5000 10000 20000 40000; do
echo "== PF_WGRAD_SKEW=$v" >> gpurun_out/wgrad_skew.log
PF_WGRAD_SKEW=$v timeout 300 python tools/microbench_train_ops.py 2>&1 | grep "^wgrad\|^# weight" >> gpurun_out/wgrad_skew.log
done
python - <<'P'
import re, collections
rows = collections.OrderedDict(); cur = None; cols=[]
for l in open("gpurun_out/wgrad_skew.log"):
    if l.startswith("=="):
        cur = l.split("=")[-1].strip(); cols.append(cur); continue
    m = re.match(r"wgrad (.*?)\s+([\d.]+) us", l)
    if m: rows.setdefault(m.group(1).strip(), {})[cur] = float(m.group(2))
    if l.startswith("# weight gradients per"): print(cur, l.strip())
print("%-28s %s" % ("layer", "  ".join("%6s" % k for k in cols)))
for k, v in rows.items():
    print("%-28s %s" % (k, "  ".join("%6.1f" % v.get(c, -1) for c in cols)))
P
