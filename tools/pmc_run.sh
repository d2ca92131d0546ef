#!/bin/bash
# rocprofv3 PMC pass over one small profile run (counters in their own run, kernel-trace only).
# usage: tools/pmc_run.sh <outdir-name> "<COUNTER LIST>"
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT -- python3 $ROOT/tools/layer_profile.py --passes 1 > $OUT.log 2>&1
