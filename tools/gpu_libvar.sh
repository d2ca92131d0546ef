#!/bin/bash
# bench.py with alternative builds of the library (MMC_LIBRARY): compiler-flag experiments
O=gpurun_out/${1:-libvar}; mkdir -p $O
run() { MMC_LIBRARY=$2 python bench.py --no-cpu-baseline 2> $O/$1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), 'patches/s', round(d['ms_per_step'],3), 'ms')" | tee -a $O/libvar.log; }
run default mermaid_classifier_amd/libmermaid_mi355.so
for f in build_variants/*.so; do
  n=$(basename $f .so)
  MMC_LIBRARY=$f python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or batch" > $O/$n.pytest 2>&1; tail -1 $O/$n.pytest
  run $n $f
done
run default2 mermaid_classifier_amd/libmermaid_mi355.so
