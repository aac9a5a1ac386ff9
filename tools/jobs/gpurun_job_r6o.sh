# round-6 job o: scene lanes in flight re-tuned after the occupancy change (PF_RESOLVE_BATCH 5)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/lanes_ab.log
for rep in 1 2; do for l in 4 3 5 6 8; do
echo "== lanes $l (rep $rep)" >> gpurun_out/lanes_ab.log
timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 --no-train-block --no-extras --lanes $l 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_depth_map'],4))" >> gpurun_out/lanes_ab.log 2>&1
done; done
cat gpurun_out/lanes_ab.log
