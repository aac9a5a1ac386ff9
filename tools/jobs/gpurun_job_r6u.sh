# round-6 job u: the kernel trace of the TIMED execution mode only (--no-extras: the one-lane / PCIe arms run other launch
# modes in the same process), folded with the PMC passes already committed
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 4 --warmup 2 --calibration-steps 2 --no-cpu-baseline --no-train-block --no-extras > $R/gpurun_out/rocprof.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/per_kernel_roofline.py summarize $DB gpurun_out/kernel_trace_cfg2_lanes4.json
rm -rf gpurun_out/prof
python tools/per_kernel_roofline.py report gpurun_out/kernel_trace_cfg2_lanes4.json gpurun_out/per_kernel_roofline --config cfg2 --pmc-fetch profiles/r06_pmc_fetch.json --pmc-write profiles/r06_pmc_write.json > /dev/null
head -40 gpurun_out/per_kernel_roofline.md | cut -c1-180
