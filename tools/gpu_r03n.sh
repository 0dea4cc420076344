#!/bin/bash
# round 3: 2-rank rehearsal of the gather with contiguous shared destinations (and the chunked form for comparison)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
for shared in 1 0; do for mode in all root; do
  PBH_GATHER_SHARED=$shared PBH_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2952$((RANDOM%10)) \
     bench.py --gpus 2 --gather $mode --no-cpu --no-series --no-extras > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("shared=$shared gather=$mode:", round(d["ms_per_step"],3), "ms/step;", d["step_with_gather_ms"])
PY
done; done
