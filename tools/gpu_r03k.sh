#!/bin/bash
# round 3: whole suite again (after the later kernel changes) + the randomised checks
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
rc=$?; tail -4 $O/suite.log
[ $rc -eq 0 ] || exit $rc
for f in "fuzz_round3.py 150" "fuzz_parity.py 120 7" "fuzz_guard.py 100" "fuzz_stft.py 60" "fuzz_fft_decode.py 60" "fuzz_stream_raw.py 60"; do
  set -- $f
  timeout -k 10 400 python tests/tools/$@ > $O/$1.log 2>&1; echo "$1 rc=$? : $(tail -1 $O/$1.log)"
done
