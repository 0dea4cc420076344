#!/bin/bash
# round 3: how should N1 of a two-level 7-smooth plan be split into P x Q?  (PBH_MIX_Q forces Q)
set -u
for q in 0 405 243 135 81 45 27; do
  echo "== 9953280 = 2^13 * 1215, PBH_MIX_Q=$q"
  PBH_MIX_Q=$q timeout -k 10 120 python tools/bench_smooth.py 9953280 2>&1 | grep "^{" | cut -c60-400
done
for q in 0 625 125 3125; do
  echo "== 12500000 = 2^5 * 390625, PBH_MIX_Q=$q"
  PBH_MIX_Q=$q timeout -k 10 120 python tools/bench_smooth.py 12500000 2>&1 | grep "^{" | cut -c60-400
done
for q in 0 625 125 25 3125; do
  echo "== 10000000 = 2^7 * 78125, PBH_MIX_Q=$q"
  PBH_MIX_Q=$q timeout -k 10 120 python tools/bench_smooth.py 10000000 2>&1 | grep "^{" | cut -c60-400
done
for q in 0 125 625 25; do
  echo "== 16000000 = 2^10 * 15625, PBH_MIX_Q=$q"
  PBH_MIX_Q=$q timeout -k 10 120 python tools/bench_smooth.py 16000000 2>&1 | grep "^{" | cut -c60-400
done
true
