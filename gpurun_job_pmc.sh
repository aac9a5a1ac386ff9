cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python tools/microbench_conv_small.py > gpurun_out/mb.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/pmc_a -o a -- python $R/tools/microbench_conv_small.py > $R/gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_b -o b -- python $R/tools/microbench_conv_small.py > $R/gpurun_out/pmc_b.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_a/a_results.db gpurun_out/pmc_a.json
python tools/pmc_summary.py gpurun_out/pmc_b/b_results.db gpurun_out/pmc_b.json
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
cat gpurun_out/mb.log; tail -3 gpurun_out/pmc_a.log
