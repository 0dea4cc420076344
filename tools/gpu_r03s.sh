#!/bin/bash
# round 3: counters of the 7-smooth column kernel (k_colmix) at 10^7 x 16 (P = 125, Q = 625) and 10 240 000 (Q = 625, one level)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_mix
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_smooth.py 10000000 9953280 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/pmc_sq2 -- python3 tools/bench_smooth.py 10000000 9953280 > $OUT/pmc_sq2.log 2>&1
python3 - <<PY
import csv, glob, collections, re
for sub in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = re.sub(r"(void )?pbh(32|64)?::", "", r["Kernel_Name"].split("(")[0])
            d[(k, r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if k[0].startswith("k_colmix") or k[0].startswith("k_radix"):
                print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
