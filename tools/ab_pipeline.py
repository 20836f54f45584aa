#!/usr/bin/env python3
"""A/B of Pipeline variants on one GPU in one process (interleaved rounds, per-kernel HIP-event times).
usage: ab_pipeline.py [n]   variants: posting width, overlapped cosine schedule"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _hip.Context(0)
lut = alphabet.build_lut("red6")
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
batch = engine.SeqBatch(ctx, res, off)
variants = {"three calls": (engine.Pipeline(ctx, lut, 12, fused=False), {}),
            "fused vectorize": (engine.Pipeline(ctx, lut, 12, fused=True), {}),
            "fused + overlap": (engine.Pipeline(ctx, lut, 12, fused=True), {"SKM_COSINE_OVERLAP": "1"}),
            "three calls, post32": (engine.Pipeline(ctx, lut, 12, post32=True, fused=False), {}),
            "three calls + overlap": (engine.Pipeline(ctx, lut, 12, fused=False), {"SKM_COSINE_OVERLAP": "1"}),
            "three calls, post32 + overlap": (engine.Pipeline(ctx, lut, 12, post32=True, fused=False), {"SKM_COSINE_OVERLAP": "1"})}
KNOBS = ("SKM_COSINE_OVERLAP",)


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)


out = None
for p, env in variants.values():
    setenv(env)
    p.out = out
    p.step(batch)
    out = p.out  # share the result buffer
ctx.sync()
ctx.profile_enable(True)
import time

rows = {}
for rnd in range(ROUNDS):
    for name, (p, env) in variants.items():
        setenv(env)
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            p.step(batch)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        prof = ctx.profile_dump()
        rows.setdefault(name, []).append((dt, {k: v[1] / 3 for k, v in prof.items()}))
for name, v in rows.items():
    v.sort(key=lambda x: x[0])
    dt, prof = v[len(v) // 2]
    print(f"{name}: {dt:.3f} ms/step  " + " ".join(f"{k}={x:.3f}" for k, x in prof.items() if x > 0.04))
