#!/bin/bash
# round 3: in-place vs ping-pong (PBH_OOP) middle passes: bench A/B + parity at the headline size
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
for m in 0 1; do
  PBH_OOP=$m timeout -k 10 300 python bench.py --no-cpu --no-series > $O/bench_oop$m.json 2> $O/bench_oop$m.err || exit 1
  python - <<PY
import json
d=json.load(open("$O/bench_oop$m.json"))
print("PBH_OOP=$m", round(d["ms_per_step"],4), d["path_roofline"]["kernel_ms"])
PY
done
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_sharded.py -k "config1 or config2 or config4 or two_shards" > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; exit $rc
