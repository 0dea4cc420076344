#!/bin/bash
# N fresh bench processes with the library's allocation-class decisions on stderr
for i in $(seq 1 ${1:-6}); do
  echo "== process $i"
  PBH_TRACE_ALLOC=1 python bench.py --no-extras --no-cpu --no-series 2> /tmp/trace_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['path_roofline']['kernel_ms'])"
  grep "four-pass roles" /tmp/trace_$i.err
  grep "pbhip\] work" /tmp/trace_$i.err
done
