#!/usr/bin/env python3
"""Distribution of exact neighbour-list lengths (rows with a non-zero dot product) for the bench workload."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
name = sys.argv[2] if len(sys.argv) > 2 else "red6"
ctx = _hip.Context(0)
lut = alphabet.build_lut(name)
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
pipe = engine.Pipeline(ctx, lut, 12)
csr = pipe.vectorize(engine.SeqBatch(ctx, res, off))
b = pipe.basis
nb = engine.gram_neighbors(ctx, csr, pipe.rnorm, n, b.ncols, b.colptr, b.post, pipe.rnorm, post_bits=b.post_bits, postcnt=b.postcnt)
ln = nb.length.download(n).astype(np.int64)
ln = ln[ln != 0xFFFFFFFF]
nnz_row = np.diff(csr.rowptr.download(n + 1))
print(f"rows {n}  entries {nb.total}  overflow rows {nb.overflow_rows}  mean {ln.mean():.1f}  max {ln.max()}")
for t in (256, 512, 1024, 2048, 4096, 8192, 16384):
    print(f"  rows with more than {t:5d} neighbours: {(ln > t).sum()}")
print(f"  rows with more than 512 non-zeros: {(nnz_row > 512).sum()}")
