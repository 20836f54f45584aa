#!/usr/bin/env python3
"""tools only: the sequence three fuzz runs of round 5 stopped in, again and again: a stream of two small batches through
engine.OverlappedPipeline on side contexts that are closed afterwards (what tests/fuzz_parity.py does every eighth round), a
one-stream step, then score.connection_matrix_from_features on small 0/1 matrices (the stops were two rounds behind the
OverlappedPipeline round, inside score._set_measure).  Watchdog and call trace as in tools/fuzz_watch.py.
usage: stress_overlap_then_score.py <iterations> [limit_s]   STRESS_NO_CLOSE=1: leave the side contexts to the garbage collector;
STRESS_NO_OVERLAP=1: skip the OverlappedPipeline part; FUZZ_SYNC=1: wait for the device after every library call"""
import collections
import faulthandler
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snekmer_amd as skm  # noqa: E402
from snekmer_amd import _hip, alphabet, engine  # noqa: E402
from snekmer_amd.synth import synth_families  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
_hip.CALL_TRACE = collections.deque(maxlen=30)
_hip.CALL_SYNC = os.environ.get("FUZZ_SYNC") == "1"
beat = [time.monotonic(), 0]


def watchdog():
    while True:
        time.sleep(2.0)
        if time.monotonic() - beat[0] > limit:
            print(f"no progress for {limit:.0f} s in iteration {beat[1]}; the last library calls (oldest first):", flush=True)
            for name, cid, ints in list(_hip.CALL_TRACE):
                print(f"  ctx {cid % 100000:5d} {name} {[v for v in ints if v is not None][:10]}", flush=True)
            faulthandler.dump_traceback(all_threads=True)
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
alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
ctx = _hip.default_context()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
names = ["hydro", "standard", "solvacc", "red6"]
t0 = time.perf_counter()
for it in range(iters):
    beat[0], beat[1] = time.monotonic(), it
    lut = alphabet.build_lut(names[it % len(names)])
    k = int(rng.integers(2, 13))
    n = int(rng.integers(20, 400))
    res, off, _ = synth_families(n, int(rng.integers(60, 600)), family=int(rng.integers(2, 40)), seed=int(rng.integers(1, 1 << 30)))
    batch = engine.SeqBatch(ctx, res, off)
    ref = engine.Pipeline(ctx, lut, k, dense_route=False)
    S = ref.step(batch)
    S = S.download().reshape(S.shape)[:n, :n].copy()
    if os.environ.get("STRESS_NO_OVERLAP") != "1":
        op = engine.OverlappedPipeline(ctx, lut, k, side_list_fraction=float(rng.choice([0.25, 0.6, 1.0])))
        op.SPLIT_MIN_ROWS = 1
        op.prefetch(batch)
        for nxt in (batch, None):
            got = op.step(nxt)
            op.sync()
            assert (got.download().reshape(got.shape)[:n, :n] == S).all(), it
        if os.environ.get("STRESS_NO_CLOSE") != "1":
            op.close()
        op = None
    for _ in range(3):
        m = int(rng.integers(2, 300))
        K = int(rng.choice([1, 7, 64, 200, 1024]))
        X = rng.random((m, K)) < float(rng.choice([0.01, 0.1, 0.6]))
        skm.score.connection_matrix_from_features(X)
        skm.score.connection_matrix_from_features(X.astype(np.float64) * 1.5, metric="cosine")
    if it % 500 == 0:
        print(f"iteration {it}: {time.perf_counter() - t0:.0f} s", flush=True)
print(f"stress ok: {iters} iterations in {time.perf_counter() - t0:.0f} s; pool {_hip.default_context().mem_stats()}", flush=True)
