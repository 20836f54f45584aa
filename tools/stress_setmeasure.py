#!/usr/bin/env python3
"""tools only: the call that three fuzz runs of round 5 stopped in - score.connection_matrix_from_features on a small 0/1 matrix
(score._set_measure: skm_dense_to_csr, skm_csr_transpose, skm_cosine_csr with norms of one, skm_setsim_f64) - again and again, with
the watchdog of tools/fuzz_watch.py: the last library calls are printed when an iteration makes no progress.
usage: stress_setmeasure.py <iterations> [limit_s]      FUZZ_SYNC=1: wait for the device after every library call"""
import collections
import faulthandler
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snekmer_amd as skm  # noqa: E402
from snekmer_amd import _hip  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
_hip.CALL_TRACE = collections.deque(maxlen=24)
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
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
t0 = time.perf_counter()
for it in range(iters):
    beat[0], beat[1] = time.monotonic(), it
    n = int(rng.integers(2, 300))
    K = int(rng.choice([1, 7, 64, 200, 1024]))
    dens = float(rng.choice([0.01, 0.1, 0.6]))
    X = rng.random((n, K)) < dens
    H = skm.score.connection_matrix_from_features(X)
    if it % 64 == 0:
        want = 1.0 - (X[:, None, :] != X[None, :, :]).mean(axis=2) if n <= 64 else None
        if want is not None:
            assert np.abs(H - want).max() <= 1e-15, it
    if it % 5000 == 0:
        print(f"iteration {it}: {time.perf_counter() - t0:.0f} s", flush=True)
print(f"stress ok: {iters} iterations in {time.perf_counter() - t0:.0f} s; pool {_hip.default_context().mem_stats()}", flush=True)
