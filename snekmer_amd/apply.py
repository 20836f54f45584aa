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


class ColumnTotals:
    """The family-total matrix by column (skm_group_postings): colptr uint32[ncols+1], post uint64 = family | total << 32
    (families ascending inside a column), normsq uint64[families] (exact squared norms of the total rows).  The layout
    skm_apply_top2 reads; `group_sum` hangs it on the CSR it returns so that `apply_top2` does not sort the totals again."""

    def __init__(self, n, ncols, nnz, colptr, post, normsq):
        self.n, self.ncols, self.nnz, self.colptr, self.post, self.normsq = n, ncols, nnz, colptr, post, normsq


def group_totals(ctx, csr: engine.CountsCSR, groups: Sequence[int], ngroups: int, basis=None, ncols: Optional[int] = None) -> ColumnTotals:
    """Per-annotation sums of the count rows (snekmer/rules/learn.smk:385-408) BY COLUMN.  `basis`: the engine.Basis of
    `csr` with its postings (engine.build_basis / the vectorize stage leave the count matrix by column: no sort is needed
    then); without it the CSR is transposed first (skm_csr_transpose; `ncols` or one look at the column ids)."""
    if csr.colidx is None:
        raise ValueError("group_sum needs column ids: call engine.build_basis first")
    if getattr(csr, "elided", False):
        raise ValueError("group_sum needs real column ids: build the basis without elide_singletons "
                         "(0xFFFFFFFF marks k-mers of one row only and is not a column)")
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    if g.size != csr.n:
        raise ValueError("one group id per row is required")
    if g.size and int(g.max()) >= ngroups:
        raise ValueError("group id out of range")
    d_g = ctx.to_device(g if g.size else np.zeros(1, np.uint32))
    nnz = csr.nnz
    if basis is not None and basis.colptr is not None and basis.post is not None and basis.post_bits == 64:
        ncols, colptr, post = basis.ncols, basis.colptr, basis.post
    else:
        if ncols is None:
            if basis is not None:
                ncols = basis.ncols
            else:
                ncols = (int(csr.colidx.download(nnz).max()) + 1) if nnz else 0
        colptr, post = engine.transpose(ctx, csr.n, nnz, ncols, csr.rowptr, csr.colidx, csr.counts)
    out_colptr = ctx.empty(ncols + 1, np.uint32)
    out_post = ctx.empty(max(nnz, 1), np.uint64)
    normsq = ctx.empty(max(ngroups, 1), np.uint64)
    got = _i64(0)
    ctx.call("skm_group_postings", _i64(csr.n), _i64(ncols), _p(colptr.ptr), _p(post.ptr), _i64(nnz), _p(d_g.ptr), _i64(ngroups),
             _p(out_colptr.ptr), _p(out_post.ptr), _p(normsq.ptr), _p(None), C.byref(got))
    return ColumnTotals(ngroups, ncols, int(got.value), out_colptr, out_post, normsq)


def group_sum(ctx, csr: engine.CountsCSR, groups: Sequence[int], ngroups: int, basis=None, ncols: Optional[int] = None) -> engine.CountsCSR:
    """Sum the count rows of each group.  `csr.colidx` must be set (engine.build_basis); the
    result is a CountsCSR over the same column space whose `colidx` (ascending per row) and
    `counts` are filled and whose `codes` are unused.  Its `columns` attribute holds the same totals by column
    (ColumnTotals), which `apply_top2` uses as they are.  `basis` (with postings): see `group_totals`."""
    cols = group_totals(ctx, csr, groups, ngroups, basis=basis, ncols=ncols)
    cap = max(cols.nnz, 1)
    out = engine.CountsCSR(ctx, ngroups, cols.nnz, 32, ctx.empty(ngroups + 1, np.int64), ctx.empty(1, np.uint32),
                           ctx.empty(cap, np.uint32), None)
    out.colidx = ctx.empty(cap, np.uint32)
    ctx.call("skm_postings_to_csr", _i64(cols.ncols), _i64(cols.nnz), _p(cols.colptr.ptr), _p(cols.post.ptr), _i64(ngroups),
             _p(out.rowptr.ptr), _p(out.colidx.ptr), _p(out.counts.ptr))
    out.columns = cols
    return out


def cosine_rows_vs_totals(ctx, csr: engine.CountsCSR, ncols: int, totals: engine.CountsCSR, mode: int = 0):
    """cosine_similarity(totals, counts).T of the reference: float32 [csr.n x totals.n] on the
    device, both operands over the same `ncols` columns."""
    if getattr(csr, "elided", False) or getattr(totals, "elided", False):
        raise ValueError("rectangular cosine needs real column ids on both sides (no elide_singletons)")
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


def apply_top2(ctx, csr: engine.CountsCSR, ncols: int, totals: engine.CountsCSR,
               row0: int = 0, row1: Optional[int] = None, order: Optional[Sequence[int]] = None):
    """The apply epilogue fused with the cosine (skm_apply_top2): for every query row the two
    best-scoring rows of `totals`, without materialising the N x A score block of
    rules/apply.smk:282-289.  Returns (idx uint32[rows,2], score float64[rows,2], dot int64[rows,2]);
    scores are formed in float64 from the exact integer dot products and squared norms, so
    ``round(score[:,0] - score[:,1], 2)`` is the reference's ``delta`` (apply.smk:320-325).
    `order`: a permutation of the rows [0, row1 - row0) in which to PROCESS them (results stay in row order): rows that
    share k-mers next to each other - e.g. ``np.argsort(labels, kind="stable")`` in learn.smk's self-evaluation - make the
    kernel's one random access per entry hit in L2."""
    row1 = csr.n if row1 is None else row1
    rows = row1 - row0
    d_order = None
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        if order.size != rows or (rows and not (np.sort(order) == np.arange(rows, dtype=np.uint32)).all()):
            raise ValueError("order must be a permutation of the rows")
        d_order = ctx.to_device(order if rows else np.zeros(1, np.uint32))
    if getattr(csr, "elided", False) or getattr(totals, "elided", False):
        raise ValueError("apply_top2 needs real column ids on both sides (no elide_singletons)")
    xsq = engine.row_normsq(ctx, csr.n, csr.rowptr, csr.counts)
    cols = totals if isinstance(totals, ColumnTotals) else getattr(totals, "columns", None)
    if cols is not None and cols.ncols == ncols:  # the totals by column already (group_sum / group_totals): nothing to sort
        colptr, post, ysq, m = cols.colptr, cols.post, cols.normsq, cols.n
    else:
        ysq = engine.row_normsq(ctx, totals.n, totals.rowptr, totals.counts)
        colptr, post = engine.transpose(ctx, totals.n, totals.nnz, ncols, totals.rowptr, totals.colidx, totals.counts)
        m = totals.n
    idx = ctx.empty(max(2 * rows, 1), np.uint32)
    score = ctx.empty(max(2 * rows, 1), np.float64)
    dot = ctx.empty(max(2 * rows, 1), np.int64)
    ctx.call("skm_apply_top2", _i64(csr.n), _p(csr.rowptr.ptr), _p(csr.colidx.ptr), _p(csr.counts.ptr), _p(xsq.ptr),
             _i64(m), _i64(ncols), _p(colptr.ptr), _p(post.ptr), _p(ysq.ptr), _i64(row0), _i64(row1),
             _p(d_order.ptr if d_order is not None else None), _p(idx.ptr), _p(score.ptr), _p(dot.ptr))
    return (idx.download(2 * rows).reshape(rows, 2), score.download(2 * rows).reshape(rows, 2),
            dot.download(2 * rows).reshape(rows, 2))


def confidence_lookup(delta: np.ndarray, table) -> np.ndarray:
    """``vals["delta"].map(global_confidence_scores)`` of rules/apply.smk:325-326: the confidence of
    every (already rounded) delta from the global confidence table, NaN where the table has no such
    key.  `table` is a mapping or a pandas Series {delta value -> confidence}; keys match by float
    equality, exactly as ``Series.map`` does."""
    if hasattr(table, "to_dict"):
        table = table.to_dict()
    keys = {float(k): float(v) for k, v in dict(table).items()}
    return np.asarray([keys.get(float(d), np.nan) for d in np.asarray(delta, dtype=np.float64)], dtype=np.float64)


def predict(ctx, csr: engine.CountsCSR, ncols: int, totals: engine.CountsCSR, labels: Optional[Sequence] = None,
            confidence=None):
    """rules/apply.smk:312-328 for a batch: Prediction (label of the best family), Score (its cosine),
    delta = round(top1 - top2, 2) and, given the global confidence table, Confidence."""
    idx, score, dot = apply_top2(ctx, csr, ncols, totals)
    top, second = score[:, 0], score[:, 1]
    delta = np.round(top - second, 2)
    out = {"top2_index": idx, "top2_score": score, "top2_dot": dot, "Score": top, "delta": delta}
    if labels is not None:
        lab = np.asarray(labels, dtype=object)
        out["Prediction"] = np.asarray([str(lab[i]) for i in idx[:, 0]], dtype=object)
    if confidence is not None:
        out["Confidence"] = confidence_lookup(delta, confidence)
    return out


def learn_apply(ctx, csr: engine.CountsCSR, ncols: int, groups: Sequence[int], ngroups: int, materialize: bool = False, basis=None):
    """Self-evaluation chain of rules/learn.smk: totals per annotation, cosine of every sequence
    against every annotation, top-2 and delta (rounded to 2 decimals as learn.smk:842/apply.smk:325).
    The N x A score block is only produced when `materialize` is set (the save_apply_associations
    branch of apply.smk:298-301 writes it out)."""
    totals = group_sum(ctx, csr, groups, ngroups, basis=basis, ncols=ncols)
    # rows of one annotation next to each other: they share their family's k-mers (see apply_top2)
    idx, score, dot = apply_top2(ctx, csr, ncols, totals, order=np.argsort(np.asarray(groups), kind="stable"))
    out = {"totals": totals, "top2_index": idx, "top2_score": score, "top2_dot": dot,
           "delta": np.round(score[:, 0] - score[:, 1], 2)}
    if materialize:
        out["scores"], out["ld"] = cosine_rows_vs_totals(ctx, csr, ncols, totals)
    return out
