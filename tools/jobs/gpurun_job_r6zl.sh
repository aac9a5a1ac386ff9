# round-6 job zl: conv3d family relabelled (PF_XCD=3) on the cfg-4 training step, same-box A/B against the default build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for m in default 3; do
if [ $m = default ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$GRAFT_REPO_ROOT/tools/experiments/libpointflow_XCD$m.so; fi
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('cfg4 xcd $m', round(d['value'],1), round(d['ms_per_step'],3))"
done; done 2>&1 | tee gpurun_out/xcd_conv3d_cfg4_ab.log
