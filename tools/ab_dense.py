#!/usr/bin/env python3
"""A/B of the dense i8 MFMA cosine variants (SKM_DENSE_VARIANT) in one process, interleaved rounds.
usage: ab_dense.py [alphabet k n]   also checks every variant against the first one bit for bit"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

name = sys.argv[1] if len(sys.argv) > 1 else "hydro"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
lut = alphabet.build_lut(name)
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 6)
# (round 5 removed variants 3 - the lock-step kernel -, 4 / 5 - v4 from row-major operands - and the timing-only ablations
# of v4 / v3 from the library: their numbers are in profiles/r02_dense_mfma.json and profiles/r04_dense_mfma.json)
variants = {"7 v4 staggered, tiled operands, full": "7", "6 v4 staggered, tiled operands, symmetric": "6",
            "11 v5 1x8 waves, B in registers, full": "11", "10 v5 1x8 waves, B in registers, symmetric": "10"}
SYMMETRIC = ("6", "10")
ABL = {}
ctx = _hip.Context(0)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.DensePipeline(ctx, lut, k)
pipe.step(batch)
kdim = pipe.kdim
ref = None
rows = {v: [] for v in variants}
ctx.profile_enable(True)
for rnd in range(5):
    for vname, env in variants.items():
        _hip.set_option("SKM_DENSE_VARIANT", env)
        ctx.profile_reset()
        out = engine.cosine_dense_i8(ctx, n, n, kdim, pipe.dense, pipe.dense, pipe.rnorm, pipe.rnorm, out=pipe.out)
        ms = ctx.profile_read("k_cosine_dense_i8")[1]
        rows[vname].append(ms)
        if rnd == 0 and vname not in ABL:
            ld = out.shape[1]
            sample = np.stack([out.download(n, offset=int(r) * ld) for r in (0, 1, 255, 256, 257, n // 2, n - 257, n - 1)])
            if ref is None:
                ref = sample
            elif vname.split()[0] in SYMMETRIC:
                # (round 5: the cells below the diagonal are scaled in the rectangular launch's order: identical bits)
                assert (sample == ref).all(), f"symmetric variant {vname} differs from the rectangular launch"
            else:
                assert (sample == ref).all(), f"variant {vname} differs from the first variant"
_hip.set_option("SKM_DENSE_VARIANT", None)
for vname, v in rows.items():
    ms = sorted(v)[len(v) // 2]
    full = 2.0 * n * n * kdim
    print(f"{vname:44s} {ms:8.3f} ms   {full / ms / 1e12:6.3f} POPS by 2NMK   (min {min(v):.3f} ms)")
