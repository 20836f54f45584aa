#!/usr/bin/env python3
"""Apply epilogue (rules/apply.smk:278-328) at scale: N query sequences against A family-total rows,
fused (skm_apply_top2: no N x A block) vs unfused (skm_cosine_csr + skm_row_top2)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, apply as skm_apply, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _hip.Context(0)
lut = alphabet.build_lut("red6")
res, off, fam = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
batch = engine.SeqBatch(ctx, res, off)
csr = engine.count_csr(ctx, batch, lut, 12)
basis = engine.build_basis(ctx, csr, lut.nsym, 12, postings=False)
nfam = int(fam.max()) + 1
totals = skm_apply.group_sum(ctx, csr, fam.astype(np.uint32), nfam)
out = {"n": n, "families": nfam, "nnz": csr.nnz, "totals_nnz": totals.nnz}
for name, fn in (("fused_apply_top2", lambda: skm_apply.apply_top2(ctx, csr, basis.ncols, totals)),
                 ("unfused_cosine_block_then_top2", lambda: skm_apply.row_top2(ctx, *(lambda s, ld: (s, n, nfam, ld))(*skm_apply.cosine_rows_vs_totals(ctx, csr, basis.ncols, totals))))):
    fn()
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
    ctx.sync()
    out[name] = {"ms_per_call_incl_host": (time.perf_counter() - t0) / 3 * 1e3,
                 "kernel_ms": {k: v[1] / 3 for k, v in ctx.profile_dump().items() if v[1] / 3 > 0.005}}
    ctx.profile_enable(False)
    if name.startswith("fused"):
        idx = r[0]
        out["top1_is_own_family_frac"] = float(np.mean(idx[:, 0] == fam))
print(json.dumps(out, indent=1))
