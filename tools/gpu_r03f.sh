#!/bin/bash
# round 3: LDS conflict fixes (layout kernels' rotated image, k_rowp16's unpadded last exchange): bench + parity
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 300 python bench.py --no-cpu --no-series > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(round(d["ms_per_step"],4), d["path_roofline"]["kernel_ms"])
PY
timeout -k 10 1000 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_incoherent_pol.py tests/test_shifts.py \
   --deselect tests/test_gpu_sharded.py::test_config3_full_size_stream > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; exit $rc
