# same-box A/B of LIBRARY variants built with different -D flags: LIB_LIST="base rb10 ..." names files
# pointmvsnet_amd/build/variants/lib_<name>.so; each arm copies its file over libpointflow_hip.so and runs the bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/libab.log
V=pointmvsnet_amd/build/variants
for rep in ${REPS:-1 2}; do
for arm in $LIB_LIST; do
  cp $V/lib_$arm.so pointmvsnet_amd/libpointflow_hip.so
  for lanes in ${LANES_LIST:-4 1}; do
  echo "== $arm lanes=$lanes (rep $rep)" >> gpurun_out/libab.log
  timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 --lanes $lanes $BENCH_ARGS 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4), 'towers_us', round(d['roofline']['towers']['kernel_us_per_depth_map'],1), 'volconv_us', round(d['roofline'].get('volume_conv',{}).get('kernel_us_per_depth_map',0),1))" >> gpurun_out/libab.log 2>&1
  done
done; done
cp $V/lib_${LIB_RESTORE:-base}.so pointmvsnet_amd/libpointflow_hip.so
cat gpurun_out/libab.log
