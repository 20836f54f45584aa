#!/usr/bin/env python3
"""tools only: from a rocprofv3 rocpd database, the dispatches of the LAST `count` kernels in start order with their
offsets (us) and durations (us), plus per-kernel totals: where a small step's time goes, launch gaps included.
usage: trace_timeline.py <db> <dispatches per step> [steps to average]"""
import sqlite3
import sys

db = sys.argv[1]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
s_col = "start" if "start" in cols else [c for c in cols if "start" in c][0]
e_col = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(con.execute(f'select name, "{s_col}", "{e_col}" from kernels order by "{s_col}"'))
print(f"{len(rows)} dispatches; columns {cols}")
if not per_step:
    # guess: dispatches between two launches of the first kernel name of the tail
    tail = rows[-400:]
    first = tail[-1][0]
    idx = [i for i, r in enumerate(tail) if r[0] == first]
    per_step = idx[-1] - idx[-2] if len(idx) >= 2 else len(tail)
last = rows[-per_step:]
t0 = last[0][1]
prev_end = t0
busy = 0
for name, s, e in last:
    print(f"{(s - t0) / 1e3:9.2f} us  +{(s - prev_end) / 1e3:7.2f} gap  {(e - s) / 1e3:8.2f} us  {name[:110]}")
    prev_end = max(prev_end, e)
    busy += e - s
span = prev_end - t0
print(f"step: {per_step} dispatches, span {span / 1e3:.1f} us, kernel sum {busy / 1e3:.1f} us")
# average span over the last few steps
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
spans = []
for i in range(1, k + 1):
    seg = rows[-per_step * i: len(rows) - per_step * (i - 1)]
    if len(seg) == per_step:
        spans.append((seg[0][1], max(r[2] for r in seg)))
if len(spans) > 1:
    per = (spans[0][1] - spans[-1][0]) / len(spans)
    print(f"average step period over the last {len(spans)} steps: {per / 1e3:.1f} us")
