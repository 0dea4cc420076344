#!/bin/bash
# round 3: rocprofv3 passes over the bench command (trace + counters), then the 2-rank rehearsal of the gather (one GPU, gloo)
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || true
bash tools/prof.sh r03 > gpurun_out/prof_r03.list 2>&1
python tools/prof_summary.py gpurun_out/prof_r03 > gpurun_out/prof_r03_summary.txt 2>&1
python tools/make_traffic.py gpurun_out/prof_r03 gpurun_out/prof_r03_traffic.json > /dev/null 2>&1
find gpurun_out/prof_r03 -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r03_kernel_stats.csv \;
# keep the merge small: drop the raw per-dispatch CSVs
find gpurun_out/prof_r03 -name "*.csv" -size +2M -delete
head -40 gpurun_out/prof_r03_summary.txt
O=gpurun_out/r03h; mkdir -p $O
for mode in all root; do
  PBH_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$((RANDOM%10)) \
     bench.py --gpus 2 --gather $mode --no-cpu --no-series --no-extras > $O/bench2_$mode.json 2> $O/bench2_$mode.err || { tail -5 $O/bench2_$mode.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/bench2_$mode.json"))
print("2 ranks on one GPU, gather=$mode:", round(d["ms_per_step"],3), "ms/step;", d["step_with_gather_ms"])
PY
done
