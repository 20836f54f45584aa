#!/usr/bin/env python3
"""Block count of the overlapped cosine schedule (diagnostic library: SKM_OVERLAP_BLOCKS), whole step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine

_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsnekmer_hip_diag.so")
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
ctx = _hip.Context(0)
res, off, _ = synth_families(100000, 300, family=100, seed=BASE_SEED + 2)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.Pipeline(ctx, alphabet.build_lut("red6"), 12)
cases = [("back to back", {})] + [(f"overlap, {b} blocks", {"SKM_COSINE_OVERLAP": "1", "SKM_OVERLAP_BLOCKS": str(b)}) for b in (4, 6, 8, 12, 16)]
rows = {}
for rnd in range(4):
    for name, env in cases:
        for k in ("SKM_COSINE_OVERLAP", "SKM_OVERLAP_BLOCKS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        pipe.step(batch)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            pipe.step(batch)
        ctx.sync()
        rows.setdefault(name, []).append((time.perf_counter() - t0) / 3 * 1e3)
for name, v in rows.items():
    print(f"{name:24s} {sorted(v)[len(v) // 2]:.3f} ms/step  (min {min(v):.3f})")
