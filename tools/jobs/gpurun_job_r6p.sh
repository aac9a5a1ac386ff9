# round-6 job p: the headline route in 2 processes x L lanes on one GPU (start barrier as the route workers use)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/procs_ab.log
run() {  # $1 processes, $2 lanes
  d=$(mktemp -d); pids=""
  for i in $(seq $1); do
    timeout 300 python bench.py --no-cpu-baseline --calibration-steps 2 --steps 10 --no-train-block --no-extras --lanes $2 --sync-dir $d > $d/out.$i 2>/dev/null &
  done
  while [ $(ls $d | grep -c ready) -lt $1 ]; do sleep 0.05; done; touch $d/go; wait
  python - $d $1 $2 <<'P' >> gpurun_out/procs_ab.log
import json,sys,glob
ls=[json.loads([l for l in open(f) if l.startswith("{")][-1]) for f in glob.glob(sys.argv[1]+"/out.*")]
maps=sum(d["steps"]*d["scenes_per_step"] for d in ls); span=max(d["wall_t1"] for d in ls)-min(d["wall_t0"] for d in ls)
print("procs %s lanes %s: %.1f maps/s (per process %s)"%(sys.argv[2],sys.argv[3],maps/span,[round(d["value"],1) for d in ls]))
P
  rm -rf $d
}
for rep in 1 2; do run 1 4; run 2 2; run 2 3; run 2 4; run 3 2; done
cat gpurun_out/procs_ab.log
