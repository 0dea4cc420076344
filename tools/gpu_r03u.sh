#!/bin/bash
# round 3: k_detect_reduce per scrunch factor (one process per factor so that the stats are per factor)
set -u
export TMPDIR=/tmp
for ns in 64 1024 16384; do
  OUT=$PWD/gpurun_out/prof_ns$ns; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/prof_detect_ns.py $ns > $OUT/log.txt 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "nscrunch $ns: $(grep -h "k_detect_reduce\|ELb1" $f | awk -F'",' '{print $1}' | cut -c1-60 | tr '\n' ' ')"
  grep -h "k_detect_reduce\|colq<1024, 1, 32, true" $f | awk -F, '{print "   ", $(NF-6), "calls, avg ns", $(NF-4)}'
done
