#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_contrib.py tests/test_abi.py > $O/tests.log 2>&1
rc=$?; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_stft_dedisp.py > $O/stft_dedisp.txt 2>&1; cat $O/stft_dedisp.txt
