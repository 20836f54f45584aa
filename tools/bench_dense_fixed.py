#!/usr/bin/env python3
"""tools only: fixed cost of the int8 GEMM cosine launch: tiny K (4 stages), one tile up to the proteome's 105 tiles."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, engine

ctx = _hip.Context(0)
rng = np.random.default_rng(0)
# (a -DSKM_DIAG build can force the split count: _hip.set_option("SKM_DENSE_SPLIT", 1))
for n, kdim in ((1024, 256), (1024, 1024), (1024, 4096), (3383, 256), (3383, 1024), (3383, 6656), (8192, 256), (8192, 6656)):
    X = (rng.random((n, kdim)) < 0.05).astype(np.int8)
    d = ctx.to_device(X)
    rn = engine.row_norms_i8(ctx, n, kdim, d)
    o = ctx.empty((n, (n + 3) // 4 * 4), np.float32)
    for _ in range(5):
        engine.cosine_dense_i8(ctx, n, n, kdim, d, d, rn, rn, out=o)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        engine.cosine_dense_i8(ctx, n, n, kdim, d, d, rn, rn, out=o)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 100 * 1e3
    nt = -(-n // 256)
    print({"n": n, "kdim": kdim, "tiles": nt * (nt + 1) // 2, "stages": kdim // 64, "ms": round(ms, 4)}, flush=True)
