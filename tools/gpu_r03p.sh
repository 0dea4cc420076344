#!/bin/bash
# round 3: 4 ranks sharing ONE GPU (gloo): the gather with three peers per rank (fd exchange among four processes, one stream per destination)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
for mode in all root; do
  PBH_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2953$((RANDOM%10)) \
     bench.py --gpus 4 --gather $mode --no-cpu --no-series --no-extras > $O/b4_$mode.json 2> $O/b4_$mode.err || { tail -8 $O/b4_$mode.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/b4_$mode.json"))
print("4 ranks on one GPU, gather=$mode:", round(d["ms_per_step"],3), "ms/step;", d["step_with_gather_ms"], "value", round(d["value"]))
PY
done
