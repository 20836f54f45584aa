#!/usr/bin/env python3
"""tools only: tests/fuzz_parity.py's loop with a watchdog: if a round does not finish within `limit` seconds the Python
stack of the waiting host thread is dumped and the process exits (a device-side hang shows which library call it is in).
usage: fuzz_watch.py <first_seed> <rounds> [limit_s]
environment: FUZZ_SYNC=1 a device wait after every library call; FUZZ_PROF=1 per-kernel timing events on the default
context, so that the report of a stop names the kernel that started and did not finish (skm_debug_report); SKM_GUARD=1
(the library's own switch) canary bytes behind every array, checked when it is freed."""
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
# the last library calls (every context) for the report of a round that does not finish; FUZZ_SYNC=1: wait for the device
# after every call, so that the last entry IS the call that hangs (at the price of the overlap between calls)
import collections  # noqa: E402
import threading  # noqa: E402

_hip.CALL_TRACE = collections.deque(maxlen=40)
_hip.CALL_SYNC = os.environ.get("FUZZ_SYNC") == "1"
beat = [time.monotonic(), seed]


def watchdog():
    while True:
        time.sleep(2.0)
        if time.monotonic() - beat[0] > limit:
            print(f"no progress for {limit:.0f} s in round {beat[1]}; the last library calls (oldest first):", flush=True)
            for name, cid, ints in list(_hip.CALL_TRACE):
                print(f"  ctx {cid % 100000:5d} {name} {[v for v in ints if v is not None][:10]}", flush=True)
            faulthandler.dump_traceback(all_threads=True)
            # a second line of defence: the report below makes runtime calls of its own
            threading.Timer(60.0, lambda: os._exit(4)).start()
            try:  # which stream is busy, which timed kernel started and did not finish, what the pool holds
                print(_hip.debug_report(), flush=True)
            except Exception as exc:  # noqa: BLE001
                print(f"debug_report: {exc}", flush=True)
            try:  # is the device busy (a kernel that never ends) or idle (the host waits for something that is not coming)?
                import subprocess

                print(subprocess.run(["rocm-smi", "--showuse", "--showmemuse"], capture_output=True, text=True, timeout=20).stdout, flush=True)
            except Exception as exc:  # noqa: BLE001
                print(f"rocm-smi: {exc}", flush=True)
            os._exit(3)


threading.Thread(target=watchdog, daemon=True).start()
ctx = _hip.default_context()
PROF = os.environ.get("FUZZ_PROF") == "1"
if PROF:
    ctx.profile_enable(True)
t0 = time.perf_counter()
for s in range(seed, seed + rounds):
    beat[0], beat[1] = time.monotonic(), s
    if PROF:
        ctx.profile_reset()
    if _hip.FREE_ERRORS:
        print("skm_free reported:", _hip.FREE_ERRORS, flush=True)
        os._exit(5)
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
print(f"fuzz ok: {rounds} rounds in {time.perf_counter() - t0:.0f} s; pool {ctx.mem_stats()}", flush=True)
