"""apply: the counts -> family totals -> cosine -> top-2 chain of ``snekmer learn`` / ``snekmer apply``
on the device.

Reference code replaced (all per sequence / per basis k-mer in Python + pandas):
  per-annotation sums of count rows      snekmer/rules/learn.smk:385-408
  cosine_similarity(totals, counts).T    snekmer/rules/apply.smk:278-289, rules/learn.smk:811-829
  top-2 scores, prediction, delta        snekmer/rules/apply.smk:312-328, rules/learn.smk:831-849

The CSV/confidence-table bookkeeping around these steps stays with the caller.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import engine

_p = C.c_void_p
_i64 = C.c_int64


def group_sum(ctx, csr: engine.CountsCSR, groups: Sequence[int], ngroups: int) -> engine.CountsCSR:
    """Sum the count rows of each group.  `csr.colidx` must be set (engine.build_basis); the
    result is a CountsCSR over the same column space whose `colidx` (ascending per row) and
    `counts` are filled and whose `codes` are unused."""
    if csr.colidx is None:
        raise ValueError("group_sum needs column ids: call engine.build_basis first")
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    if g.size != csr.n:
        raise ValueError("one group id per row is required")
    if g.size and int(g.max()) >= ngroups:
        raise ValueError("group id out of range")
    d_g = ctx.to_device(g if g.size else np.zeros(1, np.uint32))
    cap = max(csr.nnz, 1)
    out = engine.CountsCSR(ctx, ngroups, 0, 32, ctx.empty(ngroups + 1, np.int64), ctx.empty(1, np.uint32),
                           ctx.empty(cap, np.uint32), None)
    out.colidx = ctx.empty(cap, np.uint32)
    nnz = _i64(0)
    ctx.call("skm_csr_group_sum", _i64(csr.n), _i64(csr.nnz), _p(csr.rowptr.ptr), _p(csr.colidx.ptr), _p(csr.counts.ptr),
             _p(d_g.ptr), _i64(ngroups), _p(out.rowptr.ptr), _p(out.colidx.ptr), _p(out.counts.ptr), C.byref(nnz))
    out.nnz = int(nnz.value)
    return out


def cosine_rows_vs_totals(ctx, csr: engine.CountsCSR, ncols: int, totals: engine.CountsCSR, mode: int = 0):
    """cosine_similarity(totals, counts).T of the reference: float32 [csr.n x totals.n] on the
    device, both operands over the same `ncols` columns."""
    xr = engine.row_norms(ctx, csr.n, csr.rowptr, csr.counts)
    yr = engine.row_norms(ctx, totals.n, totals.rowptr, totals.counts)
    colptr, post = engine.transpose(ctx, totals.n, totals.nnz, ncols, totals.rowptr, totals.colidx, totals.counts)
    ld = (totals.n + 3) // 4 * 4
    return engine.cosine_matrix(ctx, csr, xr, totals.n, ncols, colptr, post, yr, mode=mode, ld=ld), ld


def row_top2(ctx, scores, n: int, m: int, ld: int) -> Tuple[np.ndarray, np.ndarray]:
    """(indices [n,2] uint32, values [n,2] float32): np.argsort(-S, axis=1)[:, :2] and the scores
    there, ties towards the lower column."""
    idx = ctx.empty(max(2 * n, 1), np.uint32)
    val = ctx.empty(max(2 * n, 1), np.float32)
    ctx.call("skm_row_top2", _i64(n), _i64(m), _p(scores.ptr), _i64(ld), _p(idx.ptr), _p(val.ptr))
    return idx.download(2 * n).reshape(n, 2), val.download(2 * n).reshape(n, 2)


def learn_apply(ctx, csr: engine.CountsCSR, ncols: int, groups: Sequence[int], ngroups: int):
    """Self-evaluation chain of rules/learn.smk: totals per annotation, cosine of every sequence
    against every annotation, top-2 and delta (rounded to 2 decimals as learn.smk:842/apply.smk:325)."""
    totals = group_sum(ctx, csr, groups, ngroups)
    scores, ld = cosine_rows_vs_totals(ctx, csr, ncols, totals)
    idx, val = row_top2(ctx, scores, csr.n, ngroups, ld)
    delta = np.round(val[:, 0].astype(np.float64) - val[:, 1].astype(np.float64), 2)
    return {"totals": totals, "scores": scores, "ld": ld, "top2_index": idx, "top2_score": val, "delta": delta}
