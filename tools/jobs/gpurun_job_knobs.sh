# every documented knob at its non-default value: the path must still run and give finite depth maps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/knobs.log
for kv in PF_CONV2D_WIDE=0 PF_CONV2D_WIDE_MIN=16 PF_CONV2D_WIDE_MIN=32 PF_LAZY_BN=0 PF_UNET_BOTTOM=0 PF_FEAT_HYP=0 PF_FEAT_HYP=2 \
          PF_TOWER_CL_OUT=0 PF_VC_DUAL_BN=0 PF_VC_LAZY=1 PF_DEC_LAZY=1 PF_WIDE16_TPB=1 PF_FUSED_BN=1 PF_KNN_CODES=0 PF_FETCH_CL=0 \
          PF_CONCURRENCY=0 PF_CONCURRENCY=3 PF_FORK_MODE=3 PF_GEMM_LEGACY=1 PF_KNN_LEGACY=1 PF_CONV3D_PAIR=0; do
  echo "== $kv" >> gpurun_out/knobs.log
  env $kv timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['execution'][:20])
except Exception as e:
    print('FAILED', e)" >> gpurun_out/knobs.log
done
cat gpurun_out/knobs.log
