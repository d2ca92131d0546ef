#!/bin/bash
# PMC passes over the bench workload (separate runs per counter group; program directly after `--`).
#   bash tools/gpu_counters.sh <out-dir-under-gpurun_out>
set -o pipefail
O=gpurun_out/${1:-r02pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-passes 1"
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $CMD > $O/$name.log 2>&1; echo "$name exit $?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU &&
run sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS &&
run tcc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE
