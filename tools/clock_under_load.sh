#!/bin/bash
# Shader clock and socket power (rocm-smi) while bench.py keeps every CU busy (two lanes) / half-busy kernels alone (one lane),
# against the idle readings.   usage (GPU box): bash tools/clock_under_load.sh > gpurun_out/clock.txt
smi() { rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1 | sed 's/.*(\(.*\)).*/\1/'; }
pw() { rocm-smi --showpower 2>/dev/null | grep "(W)" | head -1 | sed 's/.*: //'; }
echo "idle: sclk $(smi), power $(pw) W"
for lanes in 2 1; do
  MMC_LANES=$lanes python bench.py --no-cpu-baseline --steps 25000 --spread-blocks 1 > /tmp/clk_bench_$lanes.json 2>/dev/null &
  pid=$!
  sleep 12
  for i in 1 2 3 4 5 6 7 8; do echo "lanes=$lanes sample $i: sclk $(smi), power $(pw) W"; sleep 1; done
  wait $pid
  python -c "import json; d=json.load(open('/tmp/clk_bench_$lanes.json')); print('lanes=$lanes bench', round(d['value']), 'patches/s,', round(d['ms_per_step'],4), 'ms/step')"
done
