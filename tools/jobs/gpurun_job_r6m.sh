# round-6 job m: the tower kernels taken apart (no patch loads / no matrix instructions / no output stores)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/tower_ablation.log
for v in "" NOLOAD NOMFMA NOSTORE; do
echo "== ${v:-full}" >> gpurun_out/tower_ablation.log
env ${v:+PF_LIB_PATH=$GRAFT_REPO_ROOT/tools/experiments/libpointflow_$v.so} timeout 300 python tools/microbench_tower_ablation.py 2>&1 | grep "^tower" >> gpurun_out/tower_ablation.log
done
python - <<'P'
import re, collections
rows = collections.OrderedDict(); cur = None; cols=[]
for l in open("gpurun_out/tower_ablation.log"):
    if l.startswith("=="):
        cur = l[3:].strip(); cols.append(cur); continue
    m = re.match(r"tower (.*?): ([\d.]+) us", l)
    if m: rows.setdefault(m.group(1).strip(), {})[cur] = float(m.group(2))
print("%-28s %s" % ("layer", "  ".join("%8s" % k for k in cols)))
for k, v in rows.items():
    print("%-28s %s" % (k, "  ".join("%8.1f" % v.get(c, -1) for c in cols)))
P
