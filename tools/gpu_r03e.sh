#!/bin/bash
# (PBH_COLQ_PITCH_TEST was a timing-only switch of that experiment; it was removed from the library afterwards -- the results are in profiles/r03_colq_pitch_oop.txt)
# TIMING experiment: column passes with a padded row pitch (results wrong), in place and ping-pong
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
for cfg in "0 0" "1 0" "0 16400" "1 16400" "1 16416" "1 16448"; do
  set -- $cfg
  PBH_OOP=$1 PBH_COLQ_PITCH_TEST=$2 timeout -k 10 300 python bench.py --no-cpu --no-series > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("OOP=$1 pitch=$2", round(d["ms_per_step"],4), d["path_roofline"]["kernel_ms"])
PY
done
