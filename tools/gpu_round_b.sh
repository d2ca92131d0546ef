#!/bin/bash
# full GPU round: whole GPU suite, bench line, rocprof kernel stats, PMC passes; summaries are made on the host afterwards
set -o pipefail
T=${1:-r02b}
O=gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -x -q -m gpu -s > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -2 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cat $O/bench.json; tail -1 $O/bench.err
python tools/layer_profile.py > $O/lp.log 2>&1
python tools/tail_phases.py 256 > $O/tp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 1 > $O/stats.log 2>&1; echo "rocprof exit $?"
bash tools/gpu_counters.sh $T/pmc
