#!/bin/bash
# round 3, final state: every fuzzer once more with fresh seeds (each a few minutes; the call stays under gpurun's limit)
set -u
O=gpurun_out/r03t; mkdir -p $O
for f in "fuzz_parity.py ${S1:-150} ${SEED:-31}" "fuzz_round3.py ${S1:-150} $((${SEED:-31}+1))" "fuzz_stft.py 120 $((${SEED:-31}+2))" "fuzz_fft_decode.py 120 $((${SEED:-31}+3))" "fuzz_round2.py 120 $((${SEED:-31}+4))" "fuzz_stream_raw.py 100 $((${SEED:-31}+5))" "fuzz_detect_colq.py 60 $((${SEED:-31}+6))"; do
  set -- $f
  echo "== $f"
  timeout -k 10 400 python tests/tools/$1 $2 $3 > $O/$1.log 2>&1
  rc=$?
  tail -2 $O/$1.log
  [ $rc -eq 0 ] || { echo "rc=$rc"; exit $rc; }
done
