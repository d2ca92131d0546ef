#!/bin/bash
# diagnostic sweeps of tail7's experiment knobs (results are WRONG with a knob on; timing only)
O=gpurun_out/${1:-tune}
mkdir -p $O
for cfg in "0 0 0 0" "0 1 0 0" "0 2 0 0" "0 3 0 0" "0 0 1 0"; do
  set -- $cfg
  MMC_T7_TUNE0=$1 MMC_T7_TUNE1=$2 MMC_T7_TUNE2=$3 MMC_T7_TUNE3=$4 python tools/tail_phases.py 256 2>&1 | tee -a $O/tp.log | grep -v amdgpu.ids
done
