#!/usr/bin/env python3
"""Mean FETCH_SIZE / WRITE_SIZE per kernel name from rocprofv3 counter CSVs (KiB -> GB; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads -- for partial-line patterns the raw figure is printed too)."""
import collections
import csv
import glob
import os
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for root in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k)
            if k not in order:
                order.append(k)
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in order:
    c = acc[k]
    f = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"])) * 1024 / 1e9
    w = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])) * 1024 / 1e9
    print(f"{k[:70]:70s} FETCH raw {f:7.3f} GB (x2 = {2 * f:7.3f})  WRITE {w:7.3f} GB   n={len(c['FETCH_SIZE'])}")
