#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (our kernels only)."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith(("kf_", "kl_", "km_", "k_", "rd", "wr")):
            continue
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(rows):
    print(name)
    for c in sorted(rows[name]):
        v = rows[name][c]
        print(f"   {c:32s} mean={sum(v)/len(v):16.1f}  n={len(v)}")
