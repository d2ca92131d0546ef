#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats, and the two PMC passes (own runs).
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/<tag>_*)
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 400 $OUT/${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 1 > $OUT/${TAG}_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-passes 1 > $OUT/${TAG}_fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-passes 1 > $OUT/${TAG}_write.log 2>&1
echo write done
