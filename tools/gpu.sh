#!/bin/bash
# The GPU-box run scripts, one entry point (run through gpurun from the repo root):
#   bash tools/gpu.sh <mode> <tag> [args]        outputs under gpurun_out/<tag>/
# modes
#   iter      quick parity subset, isolated per-launch profile, tail7 phase clocks, one bench line         [TUNES="0 1" for MMC_T7_TUNE0]
#   round     whole GPU suite, bench line, layer profile, tail phases, rocprofv3 kernel stats, then `counters`
#   counters  rocprofv3 --pmc passes over the bench workload, one run per counter group (program directly after `--`)
#             CMD_OVERRIDE="python3 tools/b4_throughput.py ..." profiles another command with the same groups
#   stats     rocprofv3 --kernel-trace --stats of the bench command only                                    [CMD_OVERRIDE as above]
#   lanes     bench.py under MMC_LANES = 1..4
#   libvar    alternative builds build_variants/*.so via MMC_LIBRARY: args = bench | lp [grep] | tp   (timing experiments; such
#             builds may compute wrong results on purpose; `bench` also runs the golden/batch parity subset on each)
#   tune      tail7 experiment knobs MMC_T7_TUNE0..3 (timing only)
set -o pipefail
MODE=${1:?mode}; TAG=${2:?tag}; shift 2
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH_CMD=${CMD_OVERRIDE:-"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-passes 1 --spread-blocks 1"}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'patches/s', round(d['ms_per_step'],3), 'ms', r['kernel'], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],4))"; }

pmc() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $BENCH_CMD > $O/$name.log 2>&1; echo "$name exit $?"
}
counters() {
  pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU &&
  pmc sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS &&
  pmc tcc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
  pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
  pmc write WRITE_SIZE
}
stats() {
  local cmd=${CMD_OVERRIDE:-"python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 1 --spread-blocks 1"}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $cmd > $O/stats.log 2>&1; echo "rocprof stats exit $?"
}

case $MODE in
iter)
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_layers.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
  [ $rc -ne 0 ] && exit $rc
  python tools/layer_profile.py 2>&1 | tee $O/lp.log | grep -v amdgpu.ids
  for t in ${TUNES:-0}; do
    MMC_T7_TUNE0=$t python tools/tail_phases.py 256 2>&1 | tee -a $O/tp.log | grep -v amdgpu.ids
    MMC_T7_TUNE0=$t python bench.py --no-cpu-baseline 2> $O/bench.err | tee $O/bench.json | line bench
  done ;;
round)
  python -m pytest tests -x -q -m gpu -s > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -2 $O/pytest.log
  [ $rc -ne 0 ] && exit $rc
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cat $O/bench.json; tail -1 $O/bench.err
  python tools/layer_profile.py > $O/lp.log 2>&1
  python tools/tail_phases.py 256 > $O/tp.log 2>&1
  stats && counters ;;
counters) counters ;;
stats) stats ;;
lanes)
  for l in 1 2 3 4; do MMC_LANES=$l python bench.py --no-cpu-baseline 2> $O/l$l.err | line "lanes $l" | tee -a $O/lanes.log; done ;;
libvar)
  what=${1:-bench}
  for f in mermaid_classifier_amd/libmermaid_mi355.so build_variants/*.so; do
    n=$(basename $f .so); echo "== $n"
    case $what in
    bench)
      MMC_LIBRARY=$f python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or batch" > $O/$n.pytest 2>&1; tail -1 $O/$n.pytest
      MMC_LIBRARY=$f python bench.py --no-cpu-baseline 2> $O/$n.err | line $n | tee -a $O/libvar.log ;;
    lp) MMC_LIBRARY=$f python tools/layer_profile.py > $O/$n.lp 2>&1; grep "${2:-sum of}" $O/$n.lp ;;
    tp) MMC_LIBRARY=$f python tools/tail_phases.py 256 2>&1 | grep -v amdgpu | tee $O/$n.tp ;;
    esac
  done ;;
tune)
  for cfg in "0 0 0 0" "0 1 0 0" "0 2 0 0" "0 3 0 0" "0 0 1 0"; do
    set -- $cfg
    MMC_T7_TUNE0=$1 MMC_T7_TUNE1=$2 MMC_T7_TUNE2=$3 MMC_T7_TUNE3=$4 python tools/tail_phases.py 256 2>&1 | tee -a $O/tp.log | grep -v amdgpu.ids
  done ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
