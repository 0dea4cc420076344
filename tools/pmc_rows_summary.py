"""Sum of HBM bytes per call from rocprofv3 counter CSVs of tools/pmc_rows.py (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes)."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
for which in sorted(os.listdir(root)):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    ncall = collections.defaultdict(int)
    for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(root, which, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"(void )?pbh(32|64)?::", "", r["Kernel_Name"]).split("(")[0]
                if not k.startswith("k_") or r["Counter_Name"] != key:
                    continue
                tot[k][key] += float(r["Counter_Value"])
                if key == "FETCH_SIZE":
                    ncall[k] += 1
    gb = lambda k: (2 * tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"]) * 1024 / 1e9
    print(f"== {which}: HBM GB per call (3 calls profiled; chirp generation once)")
    s = 0.0
    for k in sorted(tot, key=lambda k: -gb(k)):
        per = gb(k) / 3
        s += per
        print(f"   {k:48s} {per:7.3f} GB  ({ncall[k]} launches)")
    print(f"   {'total':48s} {s:7.3f} GB")
