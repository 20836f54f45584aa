#!/usr/bin/env python3
"""tools only: the int8 GEMM cosine at the bench's shape (hydro k=14 counts, N = M = 32768, K = 16384): rectangular launch
(SKM_DENSE_VARIANT=11) and symmetric launch, ms and Pop/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

nm = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = _hip.Context(0)
lut = alphabet.build_lut("hydro")
res, off, _ = synth_families(nm, 300, family=100, seed=BASE_SEED + 6)
dp = engine.DensePipeline(ctx, lut, 14)
dp.step(engine.SeqBatch(ctx, res, off))
ctx.sync()
ctx.profile_enable(True)
out = {}
for label, var in (("symmetric", None), ("rectangular", "11")):
    if var:
        _hip.set_option("SKM_DENSE_VARIANT", var)
    for _ in range(2):
        engine.cosine_dense_i8(ctx, nm, nm, dp.kdim, dp.dense, dp.dense, dp.rnorm, dp.rnorm, out=dp.out)
    ctx.profile_reset()
    for _ in range(5):
        engine.cosine_dense_i8(ctx, nm, nm, dp.kdim, dp.dense, dp.dense, dp.rnorm, dp.rnorm, out=dp.out)
    ms = ctx.profile_read("k_cosine_dense_i8")[1] / 5
    out[label] = {"ms": round(ms, 3), "POPS_full_product": round(2.0 * nm * nm * dp.kdim / (ms * 1e-3) / 1e15, 3)}
    _hip.set_option("SKM_DENSE_VARIANT", None)
print(json.dumps(out))
