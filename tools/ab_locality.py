#!/usr/bin/env python3
"""Upper bound of what row ordering can buy k_gram_sparse: the same families with members stored contiguously
(every workgroup's neighbours are its launch neighbours) against the shuffled order the bench uses."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _hip.Context(0)
lut = alphabet.build_lut("red6")
out = None
for shuffle, overlap in ((True, False), (False, False), (True, True), (False, True)):
    _hip.set_option("SKM_COSINE_OVERLAP", 1 if overlap else None)
    res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2, shuffle=shuffle)
    batch = engine.SeqBatch(ctx, res, off)
    p = engine.Pipeline(ctx, lut, 12)
    p.out = out
    p.step(batch)
    out = p.out
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        p.step(batch)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    prof = ctx.profile_dump()
    ctx.profile_enable(False)
    print(f"shuffle={shuffle} overlap={overlap}: {dt:.3f} ms/step  " + " ".join(f"{k}={v[1] / 5:.3f}" for k, v in prof.items() if v[1] / 5 > 0.04))
