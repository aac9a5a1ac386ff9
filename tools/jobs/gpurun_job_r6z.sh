# round-6 job z: what the finish pass of the two-walk EdgeConv backward spends its time on (ablation switches, wrong results by design)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for size in small big; do for arm in "1 0" "0 0" "1 1" "1 2" "1 4" "1 7"; do set -- $arm
rm -rf /tmp/prof_e
PF_EDGE_BWD_SUMS=$1 PF_EDGE_FINISH_DBG=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o e -- python tools/microbench_edge_finish.py $size > /tmp/e.log 2>&1
echo "== $size sums $1 dbg $2"
DB=$(find /tmp/prof_e -name "*.db" | head -1) python - <<'P'
import os, sqlite3
con = sqlite3.connect(os.environ["DB"])
for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'edge_bwd' in name:
        print('   %-46s calls %5d avg %8.1f' % (name.split('(anonymous namespace)::')[-1].split('(')[0][:46], calls, avg / (1000.0 if avg > 5000 else 1.0)))
P
done; done 2>&1 | tee gpurun_out/edge_finish_ablation.log
