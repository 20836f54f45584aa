#!/usr/bin/env python3
"""tools only: the int8 GEMM cosine at small N (the CI proteome's shape: N = 3383, K = 6656), back to back, per forced split
count (SKM_DENSE_SPLIT) and for a few K: where a small launch's time goes."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, engine

ctx = _hip.Context(0)
rng = np.random.default_rng(0)
out = []
shapes = [(3383, 6656), (3383, 1024), (3383, 13312), (1500, 6656), (8000, 6656)]
if len(sys.argv) > 1:  # N list, K = 6656
    shapes = [(int(a), 6656) for a in sys.argv[1:]]
for n, kdim in shapes:
    X = (rng.random((n, kdim)) < 0.05).astype(np.int8)
    d = ctx.to_device(X)
    rn = engine.row_norms_i8(ctx, n, kdim, d)
    o = ctx.empty((n, (n + 3) // 4 * 4), np.float32)
    for split in (1, 2, 3, 4, 8):
        try:  # the split count can be forced in a -DSKM_DIAG build only; the product library picks it (skm_dense.hip)
            _hip.set_option("SKM_DENSE_SPLIT", split)
        except _hip.HipError:
            if split != 1:
                continue
        for _ in range(5):
            engine.cosine_dense_i8(ctx, n, n, kdim, d, d, rn, rn, out=o)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            engine.cosine_dense_i8(ctx, n, n, kdim, d, d, rn, rn, out=o)
        ctx.sync()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        ops = n * (n + 1) * kdim
        out.append({"n": n, "kdim": kdim, "split": split, "ms": round(ms, 4), "mfma_util": round(ops / (ms * 1e-3) / 5e15, 3)})
        print(out[-1], flush=True)
