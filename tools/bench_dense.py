#!/usr/bin/env python3
"""Dense small-basis path: count scatter bandwidth and i8 MFMA cosine rate on one MI355X."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

name = sys.argv[1] if len(sys.argv) > 1 else "hydro"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
ctx = _hip.Context(0)
lut = alphabet.build_lut(name)
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 5)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.DensePipeline(ctx, lut, k)
pipe.step(batch)
cd = engine.count_dense(ctx, batch, lut, k, dtype=np.uint16)
ctx.sync()
ctx.profile_enable(True)
ctx.profile_reset()
R = 5
for _ in range(R):
    pipe.step(batch)
    engine.count_dense(ctx, batch, lut, k, dtype=np.uint16)
prof = ctx.profile_dump()
space = lut.nsym**k
ms = {kname: v[1] / v[0] for kname, v in prof.items()}
gemm_ms = ms["k_cosine_dense_i8"]
ops = 2.0 * n * n * pipe.kdim
windows = int(np.maximum(np.diff(off) - k + 1, 0).sum())
fill_bytes = n * cd.shape[1] * 2
out = {
    "alphabet": name, "k": k, "n": n, "columns": space, "kdim": pipe.kdim,
    "k_cosine_dense_i8_ms": gemm_ms, "tops": ops / gemm_ms / 1e9, "i8_peak_tops": 5000.0,
    "mfma_frac": ops / gemm_ms / 1e9 / 5000.0,
    "memset_count_dense_ms": ms.get("memset_count_dense"), "k_count_dense_ms": ms.get("k_count_dense"),
    "count_dense_fill_GBps": fill_bytes / ms["memset_count_dense"] / 1e6,
    "count_dense_atomics_per_s": windows / ms["k_count_dense"] * 1e3,
    "all_ms": ms,
}
print(json.dumps(out))
