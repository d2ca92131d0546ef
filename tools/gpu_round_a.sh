#!/bin/bash
# first GPU call of the round: full GPU test suite, bench line, rocprof kernel stats, counter list
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
cat $O/bench.json
tail -2 $O/bench.err
rocprofv3 -L > $O/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 1 > $O/stats.log 2>&1; echo "rocprof exit $?"
