# SQ counters of single kernels (two rocprofv3 --pmc passes of 8 counters over tools/sq_probe.py $PROBE)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $R/gpurun_out/sqa -o a -- python $R/tools/sq_probe.py $PROBE > $R/gpurun_out/sq_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/sqb -o b -- python $R/tools/sq_probe.py $PROBE > $R/gpurun_out/sq_b.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/sqa/a_results.db gpurun_out/sq_a.json $SUMMARY_ARGS
python tools/pmc_summary.py gpurun_out/sqb/b_results.db gpurun_out/sq_b.json $SUMMARY_ARGS
rm -rf gpurun_out/sqa gpurun_out/sqb
tail -n 3 gpurun_out/sq_a.log; tail -n 3 gpurun_out/sq_b.log
