#!/bin/bash
# round 3, final state: every fuzzer once more with fresh seeds (each a few minutes; the call stays under gpurun's limit)
set -u
O=gpurun_out/r03t; mkdir -p $O
for f in "fuzz_parity.py 150 31" "fuzz_round3.py 150 32" "fuzz_stft.py 120 33" "fuzz_fft_decode.py 120 34" "fuzz_round2.py 120 35" "fuzz_stream_raw.py 100 36" "fuzz_detect_colq.py 60 37"; do
  set -- $f
  echo "== $f"
  timeout -k 10 400 python tests/tools/$1 $2 $3 > $O/$1.log 2>&1
  rc=$?
  tail -2 $O/$1.log
  [ $rc -eq 0 ] || { echo "rc=$rc"; exit $rc; }
done
