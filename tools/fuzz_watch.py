#!/usr/bin/env python3
"""tools only: tests/fuzz_parity.py's loop with a watchdog: if a round does not finish within `limit` seconds the Python
stack of the waiting host thread is dumped and the process exits (a device-side hang shows which library call it is in).
usage: fuzz_watch.py <first_seed> <rounds> [limit_s]"""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as F  # noqa: E402
from snekmer_amd import _hip  # noqa: E402
from snekmer_amd import alphabet as A  # noqa: E402

seed, rounds = int(sys.argv[1]), int(sys.argv[2])
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 90.0
if "red6" not in A.ALPHABETS:
    A.register_alphabet("red6", A.RED6_GROUPS)
ctx = _hip.default_context()
t0 = time.perf_counter()
for s in range(seed, seed + rounds):
    faulthandler.dump_traceback_later(limit, exit=True)
    F.one_round(ctx, s, verbose=True)
    if s % 4 == 0:
        print(F.dense_round(ctx, s), flush=True)
    if s % 4 == 2:
        print(F.apply_round(ctx, s), flush=True)
    if s % 4 == 1:
        print(F.records_round(ctx, s), flush=True)
    if s % 4 == 3:
        print(F.score_round(ctx, s), flush=True)
    if s % 8 == 5:
        print(F.surface_round(ctx, s), flush=True)
    faulthandler.cancel_dump_traceback_later()
print(f"fuzz ok: {rounds} rounds in {time.perf_counter() - t0:.0f} s", flush=True)
