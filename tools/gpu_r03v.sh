#!/bin/bash
# round 3: A/B of a library variant against the product on ONE box, alternating (PBHIP_LIBRARY selects the .so)
set -u
V=${1:-tools/micro/bin/libpbhip_prio3.so}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), d['path_roofline']['kernel_ms'])"; }
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-extras --no-cpu --no-series 2>/dev/null | line product
  PBHIP_LIBRARY=$PWD/$V timeout -k 10 200 python bench.py --no-extras --no-cpu --no-series 2>/dev/null | line variant
done
