# same-box A/B: round-3 tower kernels (lib_r3wide.so, plain 8 -> 8 pack) against the current ones (lib_new.so);
# CONFIGS="cfg2 cfg3 ..." (default cfg2), LANES_LIST (default "4 1"), REPS (default "1 2 3")
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/wideab.log
V=pointmvsnet_amd/build/variants
for rep in ${REPS:-1 2 3}; do
for cfg in ${CONFIGS:-cfg2}; do
for arm in r3wide new; do
  cp $V/lib_$arm.so pointmvsnet_amd/libpointflow_hip.so
  PAIR=1; [ $arm = r3wide ] && PAIR=0
  for lanes in ${LANES_LIST:-4 1}; do
  echo "== $cfg $arm lanes=$lanes (rep $rep)" >> gpurun_out/wideab.log
  PF_WIDE_PAIR=$PAIR timeout 300 python bench.py --config $cfg --no-cpu-baseline --calibration-steps 2 --steps 10 --lanes $lanes 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4), 'towers_us', round(d['roofline']['towers']['kernel_us_per_depth_map'],1))" >> gpurun_out/wideab.log 2>&1
  done
done; done; done
cp $V/lib_new.so pointmvsnet_amd/libpointflow_hip.so
cat gpurun_out/wideab.log
