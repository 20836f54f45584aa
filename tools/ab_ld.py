#!/usr/bin/env python3
"""Writer time against the row pitch of the result (ld floats per row): the same 100 k x 100 k matrix written with
different paddings, rows in input order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _hip.Context(0)
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.Pipeline(ctx, alphabet.build_lut("red6"), 12)
pipe.vectorize(batch)
b = pipe.basis
lds = [int(x) for x in sys.argv[2:]] or [n, n + 32, n + 64, n + 96, n + 128, n + 160, n + 224, n + 352, n + 480, n + 992, n + 1056, 102400, 131072]
exact = os.environ.get("SKM_AB_LD_EXACT") == "1"  # one buffer of exactly n x ld per case instead of one big one
out = None if exact else ctx.empty((n, max(lds)), np.float32)
ctx.profile_enable(True)
for ld in lds * 2:
    if exact:
        out = None
        out = ctx.empty((n, ld), np.float32)
    ctx.profile_reset()
    for _ in range(3):
        engine.cosine_matrix(ctx, pipe.csr, pipe.rnorm, n, b.ncols_hint(), b.colptr, b.post, pipe.rnorm, out=out, ld=ld,
                             post_bits=b.post_bits, postcnt=b.postcnt)
    ms = ctx.profile_read("k_cosine_write")[1] / 3
    print(f"ld = {ld:7d} floats ({ld * 4:7d} B, pitch mod 4096 = {ld * 4 % 4096:4d}, mod 256 = {ld * 4 % 256:3d}): k_cosine_write {ms:.3f} ms = {n * n * 4 / ms / 1e9:.2f} TB/s of result", flush=True)
