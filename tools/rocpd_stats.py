#!/usr/bin/env python3
"""Kernel statistics (the `rocprofv3 --kernel-trace --stats` summary) from the rocpd database
rocprofv3 writes by default on ROCm 7: one CSV row per kernel, columns as in kernel_stats.csv."""
import csv
import math
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = {}
for name, dur in con.execute("select name, duration from kernels"):
    rows.setdefault(name, []).append(int(dur))
total = sum(sum(v) for v in rows.values()) or 1
with open(out, "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        sd = math.sqrt(sum((x - avg) ** 2 for x in v) / len(v))
        w.writerow([name, len(v), sum(v), round(avg, 3), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 3)])
