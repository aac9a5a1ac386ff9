# round-6 job zg: the EdgeConv backward kernels of an EAGER cfg-4 step (real neighbour lists, no graph) under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_e
WITH_BACKWARD=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o e -- python tools/knn_indegree.py cfg4 > /tmp/e.log 2>&1
tail -3 /tmp/e.log | cut -c1-200
DB=$(find /tmp/prof_e -name "*.db" | head -1) python - <<'P'
import os, sqlite3
con = sqlite3.connect(os.environ["DB"])
for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'edge_bwd' in name:
        print('   %-60s calls %5d avg %8.1f' % (name.replace('(anonymous namespace)::','').replace('void ','')[:60], calls, avg / (1000.0 if avg > 5000 else 1.0)))
P
