#!/usr/bin/env python3
"""tools only: one small workload, a few steps, for `rocprofv3 --kernel-trace -- python3 tools/trace_small.py <what> [steps]`.
<what>: a sequence count (synthetic families, red6 k=12) or `proteome` (UP000322080, solvacc k=8: the dense route)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
what = sys.argv[1] if len(sys.argv) > 1 else "1000"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _hip.Context(0)
if what == "proteome":
    from snekmer_amd.io import read_fasta_packed

    _, res, off = read_fasta_packed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data",
                                                 "UP000322080_2603819.fasta"))
    lut, k = alphabet.build_lut("solvacc"), 8
else:
    res, off, _ = synth_families(int(what), 300, family=100, seed=BASE_SEED + 1)
    lut, k = alphabet.build_lut("red6"), 12
batch = engine.SeqBatch(ctx, res, off)
p = engine.Pipeline(ctx, lut, k)
for _ in range(5):
    p.step(batch)
ctx.sync()
for _ in range(steps):
    p.step(batch)
ctx.sync()
