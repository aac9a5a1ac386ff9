# round-6 job zd: the EdgeConv backward passes with cold caches (a 1 GB fill between repetitions) against warm
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for size in small big; do for fl in 0 1; do
rm -rf /tmp/prof_e
MB_FLUSH=$fl timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o e -- python tools/microbench_edge_finish.py $size > /tmp/e.log 2>&1
echo "== $size flush $fl"
DB=$(find /tmp/prof_e -name "*.db" | head -1) python - <<'P'
import os, sqlite3
con = sqlite3.connect(os.environ["DB"])
for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'edge_' in name or 'gemm' in name:
        print('   %-60s calls %5d avg %8.1f' % (name.replace('(anonymous namespace)::','').replace('void ','')[:60], calls, avg / (1000.0 if avg > 5000 else 1.0)))
P
done; done 2>&1 | tee gpurun_out/edge_cold_warm.log
