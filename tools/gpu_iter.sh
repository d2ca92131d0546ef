#!/bin/bash
# development iteration on the GPU box: quick parity subset, per-launch profile (isolated kernels), tail phase counters
set -o pipefail
O=gpurun_out/${1:-iter}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_layers.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/layer_profile.py 2>&1 | tee $O/lp.log | grep "tail\|back-to-back\|sum of"
for t in ${TUNES:-0}; do
  MMC_T7_TUNE0=$t python tools/tail_phases.py 256 2>&1 | tee -a $O/tp.log | grep -v amdgpu.ids
  MMC_T7_TUNE0=$t python bench.py --no-cpu-baseline 2> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), 'patches/s', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us frac', round(d['roofline']['frac'],4))"
done
