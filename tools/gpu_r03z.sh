#!/bin/bash
# round 3, final state: whole GPU suite, smoke, the default bench line, rocprofv3 passes
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
rc=$?; tail -4 $O/suite.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(round(d["ms_per_step"],4), round(d["value"]), d["roofline"]["frac"], d["path_roofline"]["frac"], d["path_roofline"]["floor_ms"], d["path_roofline"]["kernel_ms"])
print({k: (round(v["ms_per_step"],3) if "ms_per_step" in v else round(v.get("ms_total",0),1)) for k,v in d.items() if k.startswith("configs")})
PY
bash tools/prof.sh r03z > gpurun_out/prof_r03z.list 2>&1
python tools/prof_summary.py gpurun_out/prof_r03z > gpurun_out/prof_r03z_summary.txt 2>&1
python tools/make_traffic.py gpurun_out/prof_r03z gpurun_out/prof_r03z_traffic.json > /dev/null 2>&1
find gpurun_out/prof_r03z -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r03z_kernel_stats.csv \;
find gpurun_out/prof_r03z -name "*.csv" -size +2M -delete
head -8 gpurun_out/prof_r03z_summary.txt; grep "reinterleave_p2\|rowp16\|deinterleave_p2" gpurun_out/prof_r03z_summary.txt | grep CONFLICT | cut -c1-200
