#!/bin/bash
# First contact with an 8-GPU MI355X node.  The builder's leases have ONE GPU; everything below has run with
# two / four ranks time-slicing one device and with every RCCL line at world 1, never across devices.  Run from the repo root:
#     tools/first_8gpu.sh [outdir]
# 1. the sharded GPU tests with one device per rank (world 2: real peer mappings between two DIFFERENT devices, the first
#    hipMemImportFromShareableHandle / hipIpcOpenMemHandle across devices, pitched 128-byte rows written over xGMI);
# 2. bench.py at 1 / 2 / 4 / 8 ranks, results left sharded (the SCALE-shaped record: one JSON line per N);
# 3. the same with the gather to the root and to all ranks (timed separately inside bench.py: step_with_gather_ms).
# Nothing here is part of a timed driver run; it writes JSON lines the driver's SCALE file can be compared with.
set -u -o pipefail
OUT=${1:-gpurun_out/first8}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices: $NGPU" | tee "$OUT/summary.txt"
if (( NGPU < 2 )); then echo "needs at least two GPUs" | tee -a "$OUT/summary.txt"; exit 3; fi

port=29610
run_bench() {   # N extra-args tag
  local n=$1 tag=$3
  if (( n == 1 )); then
    timeout -k 10 600 python bench.py --gpus 1 --no-extras --no-cpu $2 > "$OUT/bench_${tag}_n1.json" 2> "$OUT/bench_${tag}_n1.err"
  else
    port=$((port + 1))
    timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus "$n" --no-extras --no-cpu $2 > "$OUT/bench_${tag}_n$n.json" 2> "$OUT/bench_${tag}_n$n.err"
  fi
  local rc=$?
  echo "bench $tag N=$n rc=$rc $(tail -n 1 "$OUT/bench_${tag}_n$n.json" | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('value', round(d['value']), d['unit'], 'ms/step', round(d['ms_per_step'],3), 'gather', d.get('step_with_gather_ms'))
except Exception as e: print('no JSON line:', e)")" | tee -a "$OUT/summary.txt"
  return $rc
}

# 1. peer mappings across devices, before anything is timed
PBH_TEST_ONE_DEVICE_PER_RANK=1 timeout -k 10 1500 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > "$OUT/test_gpu_sharded.log" 2>&1
echo "tests/test_gpu_sharded.py (one device per rank): rc $? -- $(tail -n 1 "$OUT/test_gpu_sharded.log")" | tee -a "$OUT/summary.txt"

# 2. scaling, results left sharded
for n in 1 2 4 8; do (( n <= NGPU )) && { run_bench $n "" sharded || break; }; done
# 3. with the gather (producers write their slices into the destination ranks' blocks over xGMI)
for mode in root all; do
  for n in 2 4 8; do (( n <= NGPU )) && { run_bench $n "--gather $mode" "gather_$mode" || break; }; done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "bench_sharded_n*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.append({"n_gpus": d["n_gpus"], "value": d["value"], "ms_per_step": d["ms_per_step"]})
    except Exception:
        pass
if rows:
    base = min(rows, key=lambda r: r["n_gpus"])
    for r in rows:
        r["speedup_vs_n1"] = r["value"] / base["value"] * base["n_gpus"] if base["n_gpus"] == 1 else None
    json.dump({"metric": "complex Msamples/s dedispersed", "scaling": "weak", "runs": rows}, open(os.path.join(out, "SCALE_first8.json"), "w"), indent=1)
    print(json.dumps(rows))
PY
