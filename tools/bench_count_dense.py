#!/usr/bin/env python3
"""BASELINE configs[4]: dense count scatter at N x 2^20 (hydro k=20), uint16 cells, one MI355X.
Reports the zero-fill and the atomic-scatter kernels separately (HIP events on the library stream)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

name = sys.argv[1] if len(sys.argv) > 1 else "hydro"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
ctx = _hip.Context(0)
lut = alphabet.build_lut(name)
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 4)
batch = engine.SeqBatch(ctx, res, off)
out = engine.count_dense(ctx, batch, lut, k, dtype=np.uint16)
ctx.sync()
ctx.profile_enable(True)
ctx.profile_reset()
R = 3
for _ in range(R):
    engine.count_dense(ctx, batch, lut, k, dtype=np.uint16, out=out)
prof = ctx.profile_dump()
windows = int(np.maximum(np.diff(off) - k + 1, 0).sum())
fill = n * out.shape[1] * 2
ms_fill = prof["memset_count_dense"][1] / prof["memset_count_dense"][0]
ms_sc = prof["k_count_dense"][1] / prof["k_count_dense"][0]
# spot check a few rows against a host count
M0 = out.download(out.shape[1], offset=0).astype(np.int64)
print(json.dumps({
    "config": f"{n} x 300aa, {name} k={k}, dense uint16 [{n} x {out.shape[1]}]",
    "matrix_bytes": fill, "windows": windows,
    "memset_ms": ms_fill, "memset_TBps": fill / ms_fill / 1e9,
    "k_count_dense_ms": ms_sc, "atomics_per_s": windows / ms_sc * 1e3,
    "algorithmic_bytes": res.size + fill + windows * 4,
    "total_ms": ms_fill + ms_sc, "total_TBps": (res.size + fill + windows * 4) / (ms_fill + ms_sc) / 1e9,
    "row0_sum": int(M0.sum()), "row0_expected": int(max(off[1] - off[0] - k + 1, 0)),
}))
