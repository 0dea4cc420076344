#!/bin/bash
# One parametrised runner for the builder's gpurun calls (replaces the per-lease scripts of rounds 2-3).
#   gpurun --timeout T -- 'tools/gpu_job.sh <tag> <limit_s> step[:seconds] ...'
# Every step has a default time limit; the limits are SUMMED and compared with <limit_s> (the lease's --timeout minus a
# margin) BEFORE anything starts -- a call that cannot finish inside its lease is refused, not cut off half way (round 3
# lost 20 GPU-minutes to a 1600-s queue in a 1200-s lease).  Steps run in order, joined by &&: after a step fails or
# times out nothing else touches the GPU.  Logs: gpurun_out/<tag>/<step>.log.
set -u -o pipefail
TAG=${1:?tag}; LIMIT=${2:?limit seconds}; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

declare -A DEF=(
  [suite]=1000 [bench]=240 [bench_full]=420 [prof]=420 [pp4bench]=120 [placement]=240 [micro]=120 [gpu_tests_fast]=600
  [ab]=420 [adopt]=10 [spread]=300 [pmcmicro]=200 [sh]=300 [pmcbench]=200 [first8]=900 [test]=600 [py]=300 [sharded]=600
)
step_suite()      { python -m pytest tests -x -q -m gpu; }
step_gpu_tests_fast() { python -m pytest tests -x -q -m gpu -k "not full_size and not fuzz"; }
step_test()       { python -m pytest -x -q -m gpu $ARG; }                      # test:SECONDS:path::name
step_sharded()    { python -m pytest -x -q -m gpu tests/test_gpu_sharded.py; }
step_bench()      { python bench.py --no-extras --no-cpu --no-series $ARG | tee "$OUT/bench_line.json"; }
step_bench_full() { python bench.py $ARG | tee "$OUT/bench_full.json"; }
step_prof()       { tools/prof.sh "$TAG" $ARG && python tools/prof_summary.py "gpurun_out/prof_$TAG" > "$OUT/prof_summary.txt" && python tools/make_traffic.py "gpurun_out/prof_$TAG" "$OUT/traffic.json"; }
# after `prof`: make this lease's traffic.json the one bench.py reads (roofline.traffic, frac_rocprof) -- same box, same build
step_adopt()      { cp "$OUT/traffic.json" profiles/traffic.json; }
step_pp4bench()   { tools/micro/bin/pp4bench; }
step_micro()      { tools/micro/bin/$ARG; }                                     # micro:SECONDS:name
step_placement()  { PBH_TRACE_ALLOC=1 python tools/exp_placement.py $ARG; }
step_py()         { python $ARG; }                                              # py:SECONDS:script args (use , for spaces)
step_ab()         { python tools/ab_env.py $ARG; }                                  # ab:SECONDS:--rounds,3,-,NAME=VAL fresh processes
# N fresh processes of the headline bench: the per-process spread (round 3: 2.5 % on one box)
step_spread()     { for i in $(seq 1 ${ARG:-5}); do python bench.py --no-extras --no-cpu --no-series 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['ms_per_step_event_median'],4), d['path_roofline']['kernel_ms'], round(d['path_roofline'].get('copy_ceiling_GBps',0)), round(d['path_roofline'].get('rmw_ceiling_GBps',0)))" || return 1; done; }
step_first8()     { tools/first_8gpu.sh; }
step_sh()         { bash $ARG; }                                                # sh:SECONDS:script args
# one counter pass over the headline bench: pmcbench:SECONDS:COUNTER,COUNTER,...  (program directly after --; no trace domains)
step_pmcbench()   { d="$OUT/pmc_$(echo $ARG | tr -c 'A-Za-z0-9' '_' | cut -c1-60)"; rocprofv3 --pmc $ARG --output-format csv -d "$d" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-series --no-extras > "$d.log" 2>&1 || { tail -5 "$d.log"; return 1; }
                    python tools/pmc_table.py "$d"; }
# FETCH_SIZE / WRITE_SIZE of a microbenchmark's kernels (separate passes; the program directly after --): pmcmicro:SECONDS:name
step_pmcmicro()   { for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- tools/micro/bin/$ARG > "$OUT/pmc_$c.log" 2>&1 || return 1; done
                    python tools/pmc_kernels.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"; }

total=0; names=(); secs=(); args=()
for spec in "$@"; do
  IFS=: read -r name s arg <<< "$spec"
  [[ -n "${DEF[$name]:-}" ]] || { echo "unknown step $name"; exit 64; }
  s=${s:-${DEF[$name]}}
  names+=("$name"); secs+=("$s"); args+=("${arg//,/ }")
  total=$((total + s))
done
if (( total > LIMIT )); then
  echo "refused: the steps' limits add up to $total s, the lease allows $LIMIT s"; exit 65
fi
echo "steps: ${names[*]} -- limits sum to $total s of $LIMIT s"
i=0
for name in "${names[@]}"; do
  ARG=${args[$i]}
  log="$OUT/$name${ARG:+_$(echo "$ARG" | tr -c 'A-Za-z0-9._-' '_' | cut -c1-40)}.log"
  t0=$(date +%s)
  export -f step_$name 2>/dev/null
  ( export ARG OUT TAG; timeout -k 10 "${secs[$i]}" bash -c "$(declare -f step_$name); step_$name" ) > "$log" 2>&1
  rc=$?
  echo "== $name${ARG:+ [$ARG]}: rc $rc, $(( $(date +%s) - t0 )) s (limit ${secs[$i]})"
  tail -n 25 "$log"
  if (( rc != 0 )); then echo "stopping after $name (rc $rc)"; exit $rc; fi
  i=$((i + 1))
done
