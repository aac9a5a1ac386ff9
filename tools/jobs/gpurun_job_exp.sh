cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/clocks.log
for lanes in 4 1; do
  echo "== lanes=$lanes" >> gpurun_out/clocks.log
  python bench.py --no-cpu-baseline --calibration-steps 1 --steps 150 --lanes $lanes > gpurun_out/clk_bench.log 2>&1 &
  BP=$!
  for i in $(seq 1 60); do
    kill -0 $BP 2>/dev/null || break
    echo "t=$i $(/opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E 'sclk|Power \(W\)|GPU use' | sed 's/.*: //; s/clock level//' | tr '\n' ' ')" >> gpurun_out/clocks.log
    sleep 0.7
  done
  wait $BP
  grep "^{" gpurun_out/clk_bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rate', round(d['value'],1))" >> gpurun_out/clocks.log
done
grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" gpurun_out/clocks.log | cut -c1-160
