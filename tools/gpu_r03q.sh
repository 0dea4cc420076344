#!/bin/bash
# round 3: kernel trace of configs[4]'s per-GPU share (detect inside the inverse column pass)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_det
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_configs.py 5 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-90s %5s %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
