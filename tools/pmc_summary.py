#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: average counter value per kernel (substring filter)."""
import collections
import csv
import glob
import re
import sys

root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            m = re.search(r"(k_[a-z0-9_]+|radix_sort[a-z_]*|scan[a-z_]*|__amd_rocclr_[A-Za-z]+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:40]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g} sum={sum(v):.4g}")
