#!/bin/bash
# A/B: k_rowp16 with the chirp phase computed in the kernel (PBH_ROW_OTF=1) instead of read; alternating, same box; parity
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
for m in 0 1 0 1; do
  PBH_ROW_OTF=$m timeout -k 10 300 python bench.py --no-cpu --no-series --no-extras > $O/b$m.json 2> $O/b$m.err || { tail -5 $O/b$m.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/b$m.json"))
print("OTF=$m", round(d["ms_per_step"],4), d["path_roofline"]["kernel_ms"])
PY
done
PBH_ROW_OTF=1 timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_sharded.py -k "config1 or config2 or config4 or two_shards" tests/test_gpu_parity.py 2>&1 | tail -3
