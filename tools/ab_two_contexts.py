#!/usr/bin/env python3
"""Throughput of consecutive steps issued alternately on two contexts (two streams, two sets of scratch, two result
buffers): the vectorize stages of step s+1 can run beside the writer of step s.  Against the same steps on one context."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lut = alphabet.build_lut("red6")
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
ctxs = [_hip.Context(0), _hip.Context(0)]
batches = [engine.SeqBatch(c, res, off) for c in ctxs]
pipes = [engine.Pipeline(c, lut, 12) for c in ctxs]
for p, b in zip(pipes, batches):
    p.step(b)
for c in ctxs:
    c.sync()


def run(k, steps=12):
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for s in range(steps):
        pipes[s % k].step(batches[s % k])
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) / steps * 1e3


for env in ({}, {"SKM_COSINE_OVERLAP": "1"}):
    _hip.set_option("SKM_COSINE_OVERLAP", env.get("SKM_COSINE_OVERLAP"))
    for k in (1, 2, 1, 2):
        print(f"{env or 'back to back'}: {k} context(s): {run(k):.3f} ms/step")
