#!/bin/bash
# What clock and power does the chip sustain under the bench load? Samples rocm-smi twice a second while `bench.py` runs its timed
# steps (no secondary workloads, no CPU leg) -> gpurun_out/clocks_$TAG.txt
TAG=${1:-r02}
mkdir -p gpurun_out
OUT=gpurun_out/clocks_$TAG.txt
: > $OUT
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$" | head -40 >> $OUT
echo "---- under load ----" >> $OUT
timeout 300 python bench.py --steps 800 --warmup 20 --secondary 0 --cpu-seconds 0 --prof-steps 0 > gpurun_out/bench_clocks_$TAG.json 2>/dev/null &
BP=$!
for i in $(seq 1 400); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $BP
tail -c 400 gpurun_out/bench_clocks_$TAG.json
echo
grep -i sclk $OUT | tail -n +2 | awk '{print}' | cut -c1-200 | tail -40
