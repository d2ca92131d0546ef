#!/bin/bash
# bench.py under 1 / 2 / 3 / 4 lanes (same kernels): how much of a step is lane overlap
O=gpurun_out/${1:-lanes}; mkdir -p $O
for l in 1 2 3 4; do
  MMC_LANES=$l python bench.py --no-cpu-baseline 2> $O/l$l.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $l', round(d['value']), 'patches/s', round(d['ms_per_step'],3), 'ms')" | tee -a $O/lanes.log
done
