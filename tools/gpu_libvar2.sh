#!/bin/bash
# layer profile (isolated kernel times) with alternative builds of the library
O=gpurun_out/${1:-libvar}; mkdir -p $O
for f in mermaid_classifier_amd/libmermaid_mi355.so build_variants/*.so; do
  n=$(basename $f .so)
  MMC_LIBRARY=$f python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or batch" > $O/$n.pytest 2>&1; tail -1 $O/$n.pytest
  MMC_LIBRARY=$f python tools/layer_profile.py > $O/$n.lp 2>&1; echo "== $n"; grep "projse\|sum of" $O/$n.lp
done
