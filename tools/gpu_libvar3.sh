#!/bin/bash
# tail7 phase clocks with alternative builds of the library (timing experiments; results of such builds may be wrong)
O=gpurun_out/${1:-libvar}; mkdir -p $O
for f in mermaid_classifier_amd/libmermaid_mi355.so build_variants/*.so; do
  n=$(basename $f .so); echo "== $n"
  MMC_LIBRARY=$f python tools/tail_phases.py 256 2>&1 | grep -v amdgpu | tee $O/$n.tp
done
