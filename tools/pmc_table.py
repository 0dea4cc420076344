#!/usr/bin/env python3
"""Mean counter values per kernel (k_* only) from one rocprofv3 --pmc output directory."""
import collections
import csv
import glob
import os
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
for f in files[-1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"])
        k = re.sub(r"(pbh(32|64)::)", "", k)
        k = re.sub(r"\(.*$", "", k)
        if not k.startswith("k_"):
            continue
        if k not in order:
            order.append(k)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in order:
    print(f"{k[:44]:44s} " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items())))
