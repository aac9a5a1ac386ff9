# round-6 job zh: counters of the EdgeConv backward kernels in an EAGER cfg-4 step (real neighbour lists) against the
# synthetic lattice of tools/microbench_edge_finish.py: L2 hit rate, wave cycles, waits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for what in step synth; do
if [ $what = step ]; then CMD="python $R/tools/knn_indegree.py cfg4"; export WITH_BACKWARD=1; else CMD="python $R/tools/microbench_edge_finish.py small big"; fi
rm -rf /tmp/pa /tmp/pb
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace -d /tmp/pa -o a -- $CMD > /tmp/pa.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d /tmp/pb -o b -- $CMD > /tmp/pb.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pa -name "*.db" | head -1) /tmp/pa.json --by-grid > /dev/null
python $R/tools/pmc_summary.py $(find /tmp/pb -name "*.db" | head -1) /tmp/pb.json --by-grid > /dev/null
echo "== $what"
python - <<'P'
import json
for f in ('/tmp/pa.json', '/tmp/pb.json'):
    d = json.load(open(f))
    d = d.get('kernels', d)
    for name in sorted(d):
        if 'edge_bwd_inverse' in name or 'edge_bwd_reduce' in name:
            short = name.replace('(anonymous namespace)::', '').replace('void ', '')
            short = short.split('(')[0] + ' ' + name.split(' grid=')[1].split(' ')[0]
            print('  %-58s' % short[:58], ' '.join('%s=%.3g' % (k.replace('SQ_', '').replace('_sum', ''), v['mean_per_dispatch'] if isinstance(v, dict) else v) for k, v in sorted(d[name].items())))
P
done 2>&1 | tee $R/gpurun_out/edge_bwd_counters.log
