"""score: pairwise similarity entry points of the hot path.

``connection_matrix_from_features`` keeps the reference signature (snekmer/score.py:149-172);
``cosine_similarity`` is the drop-in for the ``sklearn.metrics.pairwise.cosine_similarity`` calls
at rules/apply.smk:282-284, rules/learn.smk:821-823 and rules/evaluate.smk:434-436.  Both take
count matrices (non-negative integers), which is what those call sites pass; other inputs are
rejected loudly rather than routed to a CPU path.
"""
from typing import Optional

import numpy as np

from . import engine


def _as_count_csr(ctx, X):
    """Dense or scipy-sparse non-negative integer matrix -> device CSR (column ids = X's columns)."""
    try:
        import scipy.sparse as sp
    except Exception:  # pragma: no cover
        sp = None
    if sp is not None and sp.issparse(X):
        Xc = X.tocsr()
        Xc.sum_duplicates()
        data, indices, indptr, shape = Xc.data, Xc.indices, Xc.indptr, Xc.shape
    else:
        A = np.asarray(X)
        if A.ndim != 2:
            raise ValueError("expected a 2-D feature matrix")
        rows, cols = np.nonzero(A)
        data, indices = A[rows, cols], cols
        indptr = np.zeros(A.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=A.shape[0]), out=indptr[1:])
        shape = A.shape
    data = np.asarray(data)
    if data.size and (np.any(data < 0) or np.any(data != np.floor(data))):
        raise NotImplementedError(
            "snekmer_amd cosine kernels take k-mer count matrices (non-negative integers); "
            "got non-integer or negative features"
        )
    if data.size and data.max() >= 2**28:
        raise OverflowError("counts >= 2^28 are unsupported")
    csr = engine.CountsCSR(
        ctx, int(shape[0]), int(data.size), 32, ctx.to_device(np.asarray(indptr, dtype=np.int64)),
        ctx.to_device(np.zeros(max(int(data.size), 1), dtype=np.uint32)), ctx.to_device(data.astype(np.uint32) if data.size else np.zeros(1, np.uint32)), None,
    )
    csr.colidx = ctx.to_device(np.asarray(indices, dtype=np.uint32) if data.size else np.zeros(1, np.uint32))
    return csr, int(shape[1])


DENSE_MAX_COLS = 1 << 17
DENSE_MIN_DENSITY = 0.005


def _try_dense_i8(ctx, X, Y, mode, force=False):
    """Dense ndarray inputs over a small basis with counts <= 127 go to the i8 MFMA kernel
    (a true dense GEMM); everything else (sparse inputs, huge bases, large counts) returns None
    and takes the sparse path, which is exact for any count."""
    mats = [X] if Y is None else [X, Y]
    if not all(isinstance(M, np.ndarray) and M.ndim == 2 for M in mats):
        return None
    if Y is not None and X.shape[1] != Y.shape[1]:
        raise ValueError(
            f"Incompatible dimension for X and Y matrices: X.shape[1] == {X.shape[1]} while Y.shape[1] == {Y.shape[1]}"
        )
    k = X.shape[1]
    if k == 0 or (k > DENSE_MAX_COLS and not force) or min(M.shape[0] for M in mats) == 0:
        return None
    for M in mats:
        if M.dtype == bool:
            continue
        if np.any(M < 0) or np.any(M > 127) or (M.dtype.kind == "f" and np.any(M != np.floor(M))):
            return None
    if not force and sum(np.count_nonzero(M) for M in mats) < DENSE_MIN_DENSITY * sum(M.size for M in mats):
        return None
    kdim = (k + 127) // 128 * 128

    def upload(M):
        P = np.zeros((M.shape[0], kdim), dtype=np.int8)
        P[:, :k] = M
        return ctx.to_device(P), None

    dx, _ = upload(X)
    n = X.shape[0]
    xr = _dense_row_norms(ctx, dx, n, kdim)
    if Y is None:
        dy, yr, m = dx, xr, n
    else:
        dy, _ = upload(Y)
        m = Y.shape[0]
        yr = _dense_row_norms(ctx, dy, m, kdim)
    ld = (m + 3) // 4 * 4
    out = engine.cosine_dense_i8(ctx, n, m, kdim, dx, dy, xr, yr, mode=mode, ld=ld)
    return out.download().reshape(max(n, 1), max(ld, 1))[:n, :m]


def _dense_row_norms(ctx, d_mat, n, kdim):
    """1/||row|| of an int8 matrix: the Gram kernel itself gives the exact squared norms on its
    diagonal blocks, but a CSR view is cheaper: reuse skm_row_norms_csr on a one-entry-per-cell CSR."""
    import ctypes as C

    rowptr = ctx.to_device(np.arange(n + 1, dtype=np.int64) * kdim)
    counts = ctx.empty(max(n * kdim, 1), np.uint32)
    ctx.call("skm_widen_i8_u32", C.c_int64(n * kdim), C.c_void_p(d_mat.ptr), C.c_void_p(counts.ptr))
    return engine.row_norms(ctx, n, rowptr, counts)


def cosine_similarity(X, Y=None, mode: int = 0, ctx=None, path: str = "auto") -> np.ndarray:
    """Cosine similarity between the rows of X and the rows of Y (Y=None: X with itself).
    float32 result [n_x, n_y]; exact integer dot products scaled in float32 (|err| <= ~3e-7).
    `path`: "auto" picks the i8 MFMA GEMM for dense ndarrays over a small basis with counts <= 127
    and the sparse kernels otherwise; "sparse" / "dense" force one."""
    from . import _hip

    ctx = ctx or _hip.default_context()
    if path not in ("auto", "sparse", "dense"):
        raise ValueError("path must be 'auto', 'sparse' or 'dense'")
    if path != "sparse":
        dense = _try_dense_i8(ctx, X, Y, mode, force=path == "dense")
        if dense is not None:
            return dense
        if path == "dense":
            raise ValueError("dense i8 path needs dense ndarray inputs with integer values in [0, 127]")
    x, kx = _as_count_csr(ctx, X)
    if Y is None:
        y, ky = x, kx
    else:
        y, ky = _as_count_csr(ctx, Y)
        if kx != ky:
            raise ValueError(f"Incompatible dimension for X and Y matrices: X.shape[1] == {kx} while Y.shape[1] == {ky}")
    xr = engine.row_norms(ctx, x.n, x.rowptr, x.counts)
    yr = xr if y is x else engine.row_norms(ctx, y.n, y.rowptr, y.counts)
    colptr, post = engine.transpose(ctx, y.n, y.nnz, ky, y.rowptr, y.colidx, y.counts)
    ld = (y.n + 3) // 4 * 4
    out = engine.cosine_matrix(ctx, x, xr, y.n, ky, colptr, post, yr, mode=mode, ld=ld)
    return out.download().reshape(max(x.n, 1), max(ld, 1))[: x.n, : y.n]


def connection_matrix_from_features(feature_matrix, metric="jaccard"):
    """Square similarity / distance matrix between proteins (snekmer/score.py:149-172).

    metric="cosine" returns what ``sklearn.pairwise_distances(X, metric="cosine")`` returns:
    cosine *distance*, clipped to [0, 2], with an exact-zero diagonal.
    """
    if metric == "cosine":
        return cosine_similarity(feature_matrix, None, mode=1)
    if metric == "jaccard":
        return hamming_similarity(feature_matrix)
    raise NotImplementedError(
        f"metric={metric!r}: the MI355X hot path implements 'cosine' and the reference's 'jaccard' "
        "(= 1 - hamming) branches of snekmer/score.py:166-171"
    )


def jaccard_distance(feature_matrix, ctx=None) -> np.ndarray:
    """Square Jaccard distance matrix of a binary feature matrix, what
    ``squareform(pdist(X, "jaccard"))`` gives in snekmer/scripts/cluster_cluster.py:189-190 (the
    branch used when the optional BSF package is absent)."""
    return _set_measure(feature_matrix, "jaccard", ctx)


def hamming_similarity(feature_matrix, ctx=None) -> np.ndarray:
    """What the reference's metric="jaccard" branch really computes (snekmer/score.py:166-168):
    ``1 - pairwise_distances(X, metric="hamming")`` = fraction of columns on which two rows agree.
    Implemented for the binary presence matrices that branch is used with (``vecs``); the exact
    intersection sizes come from the sparse Gram kernels with unit norms."""
    return _set_measure(feature_matrix, "hamming", ctx)


def _set_measure(feature_matrix, kind: str, ctx=None) -> np.ndarray:
    import ctypes as C

    from . import _hip

    ctx = ctx or _hip.default_context()
    A = np.asarray(feature_matrix)
    if A.ndim != 2:
        raise ValueError("expected a 2-D feature matrix")
    if A.dtype != bool and np.any((A != 0) & (A != 1)):
        raise NotImplementedError(f"the {kind} measure is implemented for binary (0/1 or bool) matrices only")
    Ab = (A != 0).astype(np.uint8)
    n, ncols = Ab.shape
    if ncols == 0:
        raise ValueError("feature matrix has no columns")
    x, _ = _as_count_csr(ctx, Ab)
    ones = ctx.to_device(np.ones(n + 4, dtype=np.float32))
    colptr, post = engine.transpose(ctx, n, x.nnz, ncols, x.rowptr, x.colidx, x.counts)
    ld = (n + 3) // 4 * 4
    out = engine.cosine_matrix(ctx, x, ones, n, ncols, colptr, post, ones, mode=0, ld=ld)  # exact |a & b|
    sizes = ctx.to_device(np.concatenate([Ab.sum(axis=1), np.zeros(4)]).astype(np.float32))
    if kind == "hamming":
        ctx.call("skm_hamming_similarity_from_gram", C.c_int64(n), C.c_int64(n), C.c_int64(ncols), C.c_void_p(sizes.ptr),
                 C.c_void_p(sizes.ptr), C.c_void_p(out.ptr), C.c_int64(ld))
    else:
        ctx.call("skm_jaccard_distance_from_gram", C.c_int64(n), C.c_int64(n), C.c_void_p(sizes.ptr), C.c_void_p(sizes.ptr),
                 C.c_void_p(out.ptr), C.c_int64(ld))
    return out.download().reshape(max(n, 1), max(ld, 1))[:n, :n]
