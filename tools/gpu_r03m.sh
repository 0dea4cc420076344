#!/bin/bash
# round 3: long randomised soak (progress lines every case batch go to the logs under gpurun_out/r03m)
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
for f in "fuzz_parity.py 240 21" "fuzz_parity.py 240 22" "fuzz_parity.py 240 23" "fuzz_round3.py 240 5" "fuzz_guard.py 200" "fuzz_round2.py 200 9" "fuzz_stft.py 120" "fuzz_fft_decode.py 120"; do
  set -- $f
  timeout -k 10 500 python tests/tools/$@ > $O/$1.$3.log 2>&1; echo "$f rc=$? : $(tail -1 $O/$1.$3.log)"
done
