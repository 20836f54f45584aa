#!/usr/bin/env python3
"""BASELINE configs[3] at one GPU's share: vectorize 1 M x 300 aa (red6 k=12) and compute the
neighbour lists + top-10 cosine neighbours for one 125 k-row block against all 1 M columns
(the dense 125 k x 1 M block would be 500 GB; the reduced output is what fits)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
block = int(sys.argv[2]) if len(sys.argv) > 2 else n // 8
alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
ctx = _hip.Context(0)
lut = alphabet.build_lut("red6")
t0 = time.perf_counter()
res, off, fam = synth_families(n, 300, family=100, seed=BASE_SEED + 3)
t_gen = time.perf_counter() - t0
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.Pipeline(ctx, lut, 12)
out = {}
for rnd in range(2):
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    pipe.vectorize(batch)
    ctx.sync()
    t_vec = time.perf_counter() - t0
    b = pipe.basis
    t0 = time.perf_counter()
    nb = engine.gram_neighbors(ctx, pipe.csr, n, b.ncols, b.colptr, b.post, row0=0, row1=block,
                               cap_entries=block * 6000)
    ctx.sync()
    t_nb = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, val = engine.neighbors_topk(ctx, nb, pipe.rnorm, pipe.rnorm, 10, exclude_self=True)
    t_top = time.perf_counter() - t0
    prof = ctx.profile_dump()
# sanity: the best neighbour of a row is a member of its family
same = float(np.mean(fam[idx[:, 0].astype(np.int64) % n] == fam[:block]))
print(json.dumps({
    "config": f"{n} x 300aa red6 k=12 on one MI355X; neighbour lists + top-10 for rows [0,{block}) x {n} columns",
    "generate_s": t_gen, "vectorize_ms": t_vec * 1e3, "neighbors_ms": t_nb * 1e3, "topk_ms_incl_download": t_top * 1e3,
    "nnz": pipe.csr.nnz, "basis_columns": b.ncols, "list_entries": nb.total, "overflow_rows": nb.overflow_rows,
    "entries_per_row": nb.total / block, "top1_same_family_frac": same,
    "sequences_per_s_vectorize": n / t_vec, "rows_per_s_neighbors": block / t_nb,
    "kernels_ms": {k: v[1] for k, v in prof.items()},
}))
