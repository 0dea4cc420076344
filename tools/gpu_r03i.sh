#!/bin/bash
# round 3: long blocks / m*2^k after the k_reint_radix rewrite (no spills), k_seg_pair addressing; parity + timings
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_contrib.py tests/test_bench_contract.py > $O/tests.log 2>&1
rc=$?; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_long.py > $O/long.jsonl 2> $O/long.err; cat $O/long.jsonl
timeout -k 10 300 python tests/tools/bench_odd.py > $O/odd.txt 2>&1; tail -12 $O/odd.txt
timeout -k 10 300 python tools/bench_stft.py > $O/stft.txt 2>&1; tail -12 $O/stft.txt
