#!/bin/bash
# isolated kernel times (layer profile) with alternative builds of the library: timing-only experiments (results of such builds may be wrong)
O=gpurun_out/${1:-libvar}; mkdir -p $O
for f in mermaid_classifier_amd/libmermaid_mi355.so build_variants/*.so; do
  n=$(basename $f .so); echo "== $n"
  MMC_LIBRARY=$f python tools/layer_profile.py > $O/$n.lp 2>&1; grep "${2:-mid14}" $O/$n.lp
done
