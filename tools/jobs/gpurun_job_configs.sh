# the other BASELINE configurations' bench lines (cfg 1 / 3 / 5), cfg 2 on one lane, and the reference's model.py on the
# drop-in layer -- the rows of DESIGN.md section 7's table that the default bench line does not cover
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in cfg1 cfg3 cfg5; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_$cfg.json
done
timeout 300 python bench.py --lanes 1 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_cfg2_lanes1.json
REF=$(ls oracle/_ref/pointmvsnet/model.py.txt 2>/dev/null || ls oracle/_ref/reference_model_py.txt 2>/dev/null)
[ -n "$REF" ] && timeout 400 python bench.py --route reference-model --reference-model-py $REF --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_route_reference_model.json
for f in gpurun_out/bench_cfg1.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg5.json gpurun_out/bench_cfg2_lanes1.json gpurun_out/bench_route_reference_model.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); r=d.get('roofline') or {}
print('$f', round(d['value'],2), d['unit'], round(d['ms_per_step'],3), 'towers', round((r.get('towers') or {}).get('frac_of_f32_mfma_peak',0),3), 'volconv', round((r.get('volume_conv') or {}).get('frac_of_f32_mfma_peak',0),3), 'gather', round((r.get('gather_path') or {}).get('frac_of_hbm_peak',0),3), round((r.get('gather_path') or {}).get('whole_step_frac',0),3))"; done
