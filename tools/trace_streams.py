#!/usr/bin/env python3
"""tools only: from a rocprofv3 rocpd database of a pipelined run (engine.OverlappedPipeline), the last steps' kernels per
hardware queue: for every queue the busy time per step and the kernels with their share; when each queue is busy relative to
the writer launches.  usage: trace_streams.py <db> [steps]"""
import collections
import sqlite3
import sys

db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
print("columns:", cols)
rows = list(con.execute(f'select name, start, end, {qcol or "0"} from kernels order by start'))
# steps: delimited by the k_cosine_write launches (two per step in the pipelined run: the second closes the step)
writes = [i for i, r in enumerate(rows) if "k_cosine_write" in r[0]]
per = 2
skip = 8  # (the bench's identity check behind the timed region launches a few writers of its own)
last = writes[-per * steps - 1 - skip:-skip]
t_begin, t_end = rows[last[0]][2], rows[last[-1]][2]
span = (t_end - t_begin) / steps
print(f"{steps} steps, {span / 1e6:.3f} ms per step (between writer completions)")
sel = [r for r in rows if r[1] >= t_begin and r[2] <= t_end]
byq = collections.defaultdict(lambda: collections.defaultdict(float))
for name, s, e, q in sel:
    import re
    m = re.search(r"(k_[a-z0-9_]+|radix_sort[a-z_]*|scan[a-z_]*|__amd_rocclr_[A-Za-z]+|onesweep::[a-z_]+)", name)
    short = m.group(1) if m else name[:40]
    byq[q][short] += (e - s)
for q, d in byq.items():
    tot = sum(d.values())
    print(f"queue {q}: busy {tot / steps / 1e6:.3f} ms per step")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:14]:
        print(f"     {v / steps / 1e6:7.3f} ms  {k}")

# one step in start order with queue ids (the last one)
one = [r for r in rows if r[1] >= rows[last[-3]][2] - 1500000 and r[2] <= rows[last[-1]][2]]
t0 = one[0][1]
print("last step, start order:")
import re
for name, s_, e_, q in one:
    m = re.search(r"(k_[a-z0-9_]+|radix_sort[a-z_]*|scan[a-z_]*|__amd_rocclr_[A-Za-z]+|onesweep::[a-z_]+)", name)
    if (e_ - s_) > 20000:
        print(f"  q{q} {(s_ - t0) / 1e6:8.3f} -> {(e_ - t0) / 1e6:8.3f} ms  {(e_ - s_) / 1e6:6.3f}  {m.group(1) if m else name[:40]}")
