#!/bin/bash
# round 3: whole GPU suite, then the default bench line (with extras and the CPU leg)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
rc=$?; tail -6 $O/suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
rc=$?; cat $O/bench_default.json; tail -8 $O/bench_default.err; exit $rc
