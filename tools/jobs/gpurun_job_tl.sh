cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lz in 0 1; do
PF_LAZY_BN=$lz PF_TIMELINE=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/tl_lazy$lz.json
python - <<PY
import json
d=json.loads(open("gpurun_out/tl_lazy$lz.json").read())
print("lazy=$lz", d["value"], d.get("stage_timeline_us"))
PY
done
