"""score: pairwise similarity entry points of the hot path.

``connection_matrix_from_features`` keeps the reference signature (snekmer/score.py:149-172);
``cosine_similarity`` is the drop-in for the ``sklearn.metrics.pairwise.cosine_similarity`` calls
at rules/apply.smk:282-284, rules/learn.smk:821-823 and rules/evaluate.smk:434-436 (ndarrays,
DataFrames or scipy sparse matrices in, float64 ndarray out, as sklearn).

Three device paths, all exact in the sense the reference needs (|error| <= 1e-5, in practice far less):
count matrices (non-negative integers, what the rule call sites pass) take the exact-integer sparse
Gram (int32 cells where the rows' norms prove that every dot product fits, float64 accumulators for the
rows where they do not: include/snekmer_hip.h, skm_cosine_csr) or, for small dense bases, the i8 MFMA GEMM,
both with float32 scaling; any other real-valued
matrix (e.g. the length-normalised rows of snekmer/utils.py:183-203) takes a float64 GEMM on the
f64 matrix cores with sklearn's own order of operations.  There is no CPU path.
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
        raise ValueError("internal: real-valued features must take the float64 path")
    if data.size and data.max() >= 2**32:  # cosine_similarity sends such matrices to the float64 path
        raise OverflowError("counts >= 2^32 do not fit the CSR's uint32 cells")
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
    xr = engine.row_norms_i8(ctx, n, kdim, dx)  # straight from the int8 operand (skm_row_norms_i8)
    if Y is None:
        dy, yr, m = dx, xr, n
    else:
        dy, _ = upload(Y)
        m = Y.shape[0]
        yr = engine.row_norms_i8(ctx, m, kdim, dy)
    ld = (m + 3) // 4 * 4
    out = engine.cosine_dense_i8(ctx, n, m, kdim, dx, dy, xr, yr, mode=mode, ld=ld)
    return out.download().reshape(max(n, 1), max(ld, 1))[:n, :m]


def _plain(M):
    """DataFrame -> ndarray (the rule call sites pass DataFrames, rules/apply.smk:282-284); scipy sparse
    and ndarrays pass through."""
    try:
        import scipy.sparse as sp

        if sp.issparse(M):
            return M
    except Exception:  # pragma: no cover
        pass
    if hasattr(M, "to_numpy"):
        return M.to_numpy()
    return np.asarray(M)


def _values(M):
    return M.data if hasattr(M, "tocsr") else M


def _is_count_matrix(M) -> bool:
    v = np.asarray(_values(M))
    if v.dtype == bool or v.size == 0:
        return True
    if v.dtype.kind in "ui":
        return bool(v.min() >= 0)
    if v.dtype.kind != "f":
        return False
    return bool(np.all(np.isfinite(v)) and v.min() >= 0 and np.all(v == np.floor(v)))


def _fits_u32(M) -> bool:
    """Counts are uint32 on the device; a count matrix with a larger cell is just a real-valued matrix (float64 path)."""
    v = np.asarray(_values(M))
    return bool(v.dtype == bool or v.size == 0 or v.max() < 2**32)


F64_MAX_ELEMS = 1 << 31  # dense float64 operands above 16 GiB are refused rather than silently densified


def _cosine_f64(ctx, X, Y, mode):
    """Real-valued features: float64 GEMM of row-normalised operands (skm_cosine_dense_f64)."""
    import ctypes as C

    def dense(M):
        if hasattr(M, "toarray"):
            if M.shape[0] * M.shape[1] > F64_MAX_ELEMS:
                raise NotImplementedError("real-valued sparse features this large are not supported: densify in blocks")
            M = M.toarray()
        A = np.ascontiguousarray(M, dtype=np.float64)
        if A.ndim != 2:
            raise ValueError("expected a 2-D feature matrix")
        if not np.all(np.isfinite(A)):
            raise ValueError("Input contains NaN or infinity.")
        return A

    A = dense(X)
    B = A if Y is None else dense(Y)
    if A.shape[1] != B.shape[1]:
        raise ValueError(
            f"Incompatible dimension for X and Y matrices: X.shape[1] == {A.shape[1]} while Y.shape[1] == {B.shape[1]}"
        )
    n, k = A.shape
    m = B.shape[0]
    if n == 0 or m == 0:
        return np.zeros((n, m), dtype=np.float64)
    dx = ctx.to_device(A if A.size else np.zeros(1))
    dy = dx if B is A else ctx.to_device(B if B.size else np.zeros(1))
    out = ctx.empty((n, max(m, 1)), np.float64)
    ctx.call("skm_cosine_dense_f64", C.c_int64(n), C.c_int64(m), C.c_int64(k), C.c_void_p(dx.ptr), C.c_int64(k),
             C.c_void_p(dy.ptr), C.c_int64(k), mode, C.c_void_p(out.ptr), C.c_int64(m))
    return out.download().reshape(n, m)


def cosine_similarity(X, Y=None, mode: int = 0, ctx=None, path: str = "auto", dtype=np.float64) -> np.ndarray:
    """Cosine similarity (mode 0) or sklearn's cosine distance (mode 1: 1 - s clipped to [0, 2], with an exact-zero
    diagonal when Y is None or Y is X, as sklearn.metrics.pairwise.cosine_distances) between the rows of X and the
    rows of Y (Y=None: X with itself), [n_x, n_y], float64 like sklearn's (pass dtype=np.float32 to keep the device's float32 block of the count
    paths without the widening copy).
    Count matrices: exact integer dot products scaled in float32 (|err| <= ~3e-7); `path` "auto" picks
    the i8 MFMA GEMM for dense ndarrays over a small basis with counts <= 127 and the sparse kernels
    otherwise, "sparse" / "dense" force one.  Any other real-valued input: float64 on the f64 matrix
    cores (path "f64"; |err| ~1e-15)."""
    from . import _hip

    ctx = ctx or _hip.default_context()
    if path not in ("auto", "sparse", "dense", "f64"):
        raise ValueError("path must be 'auto', 'sparse', 'dense' or 'f64'")
    if mode not in (0, 1):
        raise ValueError("mode must be 0 (similarity) or 1 (distance)")
    if Y is X:  # sklearn: cosine_distances zeroes the diagonal `if X is Y or Y is None`
        Y = None
    X = _plain(X)
    Y = None if Y is None else _plain(Y)
    counts = _is_count_matrix(X) and (Y is None or _is_count_matrix(Y)) and _fits_u32(X) and (Y is None or _fits_u32(Y))
    if path == "f64" or (path == "auto" and not counts):
        return _cosine_f64(ctx, X, Y, mode).astype(dtype, copy=False)
    if not counts:
        raise ValueError(f"path={path!r} takes count matrices (non-negative integers); use path='f64' or 'auto'")
    if mode == 1 and Y is not None:
        mode = 2  # distance between two different matrices: no diagonal rule (include/snekmer_hip.h)
    return _cosine_counts(ctx, X, Y, mode, path).astype(dtype, copy=False)


def _cosine_counts(ctx, X, Y, mode, path):
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
    if metric in ("hamming", "matching"):
        return _set_measure(feature_matrix, 2)
    return pairwise_distances(feature_matrix, metric=metric)


# what sklearn.metrics.pairwise_distances accepts as a metric name (scikit-learn 1.7: _VALID_METRICS + "precomputed")
SKLEARN_METRICS = ("braycurtis", "canberra", "chebyshev", "cityblock", "correlation", "cosine", "dice", "euclidean", "hamming", "haversine",
                   "jaccard", "l1", "l2", "mahalanobis", "manhattan", "matching", "minkowski", "nan_euclidean", "precomputed", "rogerstanimoto",
                   "russellrao", "seuclidean", "sokalmichener", "sokalsneath", "sqeuclidean", "wminkowski", "yule")

# metric name -> id of skm_pairwise_f64 (include/snekmer_hip.h)
PAIRWISE_METRICS = {"cityblock": 0, "manhattan": 0, "l1": 0, "sqeuclidean": 1, "euclidean": 2, "l2": 2, "chebyshev": 3, "canberra": 4,
                    "braycurtis": 5, "minkowski": 6, "nan_euclidean": 7, "haversine": 8, "dice": 10, "rogerstanimoto": 11, "russellrao": 12, "sokalmichener": 13,
                    "sokalsneath": 14, "yule": 15}


def pairwise_distances(X, metric: str = "euclidean", p: float = 2.0, ctx=None) -> np.ndarray:
    """``sklearn.metrics.pairwise_distances(X, metric=metric)`` (the `else` branch of snekmer/score.py:169-171) on the
    device for the metrics that are sums or maxima over columns (cityblock / manhattan / l1, euclidean / l2,
    sqeuclidean, chebyshev, canberra, braycurtis, minkowski) and scipy's boolean dissimilarities (dice,
    rogerstanimoto, russellrao, sokalmichener, sokalsneath, yule; the input read as x != 0); "cosine", "hamming" /
    "matching" go to their own kernels.  Square float64 matrix with an exact-zero diagonal.  "euclidean" / "l2" sum
    (x - y)^2 directly where sklearn takes sqrt(xx + yy - 2 xy): the two agree to the rounding error of sklearn's
    expansion (eps * (xx + yy) in the squared distance), and duplicate rows get an exact 0 here.
    The three metrics that need more than the two rows are one O(n d^2) pass over the matrix on the host followed by a
    device kernel on the transformed rows (`_whole_matrix_metric`): "correlation" = cosine distance of the row-centred
    matrix, "seuclidean" = euclidean after dividing every column by its standard deviation (V = var(X, ddof=1), what
    sklearn passes to scipy), "mahalanobis" = euclidean after X -> X L with L L^T = VI = inv(cov(X^T))^T.  The rest of
    scikit-learn's names raise NotImplementedError - no Snekmer rule passes them -, a name scikit-learn does not know raises
    its ValueError.

    Round 5: "nan_euclidean" (NaN entries allowed: sklearn's nan_euclidean_distances, the squared distance over the columns
    present in both rows scaled to all columns, NaN where no column is), "haversine" (exactly two columns, radians) and
    "precomputed" (the matrix itself after scikit-learn's checks: square, no negative value) are implemented too; what
    still raises NotImplementedError is "wminkowski" (gone from scipy), the true "jaccard" through this function, and
    callables (a Python function per pair of rows has no device form)."""
    import ctypes as C

    from . import _hip

    if callable(metric):
        raise NotImplementedError("a callable metric is a Python function per pair of rows: it has no device form; "
                                  "call sklearn.metrics.pairwise_distances for it")
    if metric == "precomputed":
        # sklearn: check_pairwise_arrays(precomputed=True) + check_non_negative, then the matrix itself
        A = _plain(X)
        if hasattr(A, "toarray"):
            A = A.toarray()
        A = np.asarray(A, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError(f"Precomputed metric requires shape (n_queries, n_indexed). Got {A.shape} for {A.shape[0] if A.ndim else 0} indexed.")
        if not np.all(np.isfinite(A)):
            raise ValueError("Input contains NaN or infinity.")
        if A.size and A.min() < 0:
            raise ValueError("Negative values in data passed to `pairwise_distances`. Precomputed distance  need to have non-negative values.")
        return A.copy()
    if metric == "cosine":
        return cosine_similarity(X, None, mode=1, ctx=ctx)
    if metric in ("hamming", "matching"):
        return _set_measure(X, 2, ctx)
    if metric in ("correlation", "seuclidean", "mahalanobis"):
        return _whole_matrix_metric(X, metric, ctx)
    if not isinstance(metric, str) or metric not in SKLEARN_METRICS:
        # scikit-learn's own refusal (an InvalidParameterError, which is a ValueError) for a name it does not know
        raise ValueError(f"The 'metric' parameter of pairwise_distances must be a str among {set(SKLEARN_METRICS)} or a callable. "
                         f"Got {metric!r} instead.")
    if metric not in PAIRWISE_METRICS:
        raise NotImplementedError(
            f"metric={metric!r}: implemented on the device are 'cosine', 'hamming', the reference's default 'jaccard' "
            f"(= 1 - hamming), 'correlation', 'seuclidean', 'mahalanobis' and {sorted(PAIRWISE_METRICS)} (snekmer/score.py:166-171); "
            f"no Snekmer rule passes another metric")
    ctx = ctx or _hip.default_context()
    A = _plain(X)
    if hasattr(A, "toarray"):
        raise TypeError("scipy distance metrics do not support sparse matrices.")  # sklearn's message for these metrics
    A = np.ascontiguousarray(np.asarray(A), dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("expected a 2-D feature matrix")
    if metric == "nan_euclidean":
        if np.any(np.isinf(A)):
            raise ValueError("Input contains infinity or a value too large for dtype('float64').")
    elif not np.all(np.isfinite(A)):
        raise ValueError("Input contains NaN or infinity.")
    n, k = A.shape
    if k == 0:
        raise ValueError("feature matrix has no columns")
    if metric == "haversine" and k != 2:
        raise ValueError("Haversine distance only valid in 2 dimensions")
    if n == 0:
        return np.zeros((0, 0), dtype=np.float64)
    dx = ctx.to_device(A)
    out = ctx.empty((n, n), np.float64)
    ctx.call("skm_pairwise_f64", PAIRWISE_METRICS[metric], C.c_double(p), C.c_int64(n), C.c_int64(n), C.c_int64(k), C.c_void_p(dx.ptr),
             C.c_int64(k), C.c_void_p(dx.ptr), C.c_int64(k), C.c_void_p(out.ptr), C.c_int64(n))
    return out.download().reshape(n, n)


def _whole_matrix_metric(X, metric: str, ctx=None) -> np.ndarray:
    """sklearn.metrics.pairwise_distances(X, metric=metric) for "correlation", "seuclidean" and "mahalanobis"
    (snekmer/score.py:169-171: sklearn hands these to scipy's pdist, with V / VI computed from X).  The part that
    looks at the whole matrix (row means; column variances; the inverse covariance and its Cholesky factor) is O(n d^2)
    on the host in float64; the O(n^2 d) pairwise part runs on the device on the transformed rows.  NaN where scipy
    gives NaN: a constant row has no correlation with anything, a constant column has no standardised difference."""
    A = _plain(X)
    if hasattr(A, "toarray"):
        raise TypeError("scipy distance metrics do not support sparse matrices.")
    A = np.ascontiguousarray(np.asarray(A), dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("expected a 2-D feature matrix")
    if not np.all(np.isfinite(A)):
        raise ValueError("Input contains NaN or infinity.")
    n, d = A.shape
    if d == 0:
        raise ValueError("feature matrix has no columns")
    if n < 2:
        return np.zeros((n, n), dtype=np.float64)
    if metric == "correlation":
        centred = A - A.mean(axis=1, keepdims=True)
        D = _cosine_f64(ctx or _default_ctx(), centred, None, 1)
        flat = ~np.any(centred != 0, axis=1)  # scipy: 0 / 0
        if flat.any():
            D[flat, :] = np.nan
            D[:, flat] = np.nan
            np.fill_diagonal(D, 0.0)
        return D
    if metric == "seuclidean":
        V = np.var(A, axis=0, ddof=1)
        if np.any(V == 0):  # every difference in that column is 0: 0 / 0 in each pair's sum
            D = np.full((n, n), np.nan)
            np.fill_diagonal(D, 0.0)
            return D
        return pairwise_distances(A / np.sqrt(V), metric="euclidean", ctx=ctx)
    VI = np.linalg.inv(np.atleast_2d(np.cov(A.T))).T  # sklearn's _precompute_metric_params; LinAlgError when singular
    try:
        L = np.linalg.cholesky(0.5 * (VI + VI.T))  # the quadratic form only sees the symmetric part
    except np.linalg.LinAlgError:
        raise ValueError("mahalanobis: the inverse covariance of X is not positive definite (more columns than "
                         "independent rows?); scipy's result is not a distance there") from None
    return pairwise_distances(A @ L, metric="euclidean", ctx=ctx)


def _default_ctx():
    from . import _hip

    return _hip.default_context()


def jaccard_distance(feature_matrix, ctx=None) -> np.ndarray:
    """Square Jaccard distance matrix, what ``squareform(pdist(X, "jaccard"))`` gives in
    snekmer/scripts/cluster_cluster.py:189-190 (the branch used when the optional BSF package is absent):
    |a xor b| / |a or b| on the rows' non-zero patterns (scipy reads numeric rows as booleans), 0 for two empty
    rows.  float64."""
    A = _plain(feature_matrix)
    if hasattr(A, "toarray"):
        A = A.toarray()
    A = np.asarray(A)
    if A.dtype.kind == "f" and not np.all(np.isfinite(A)):
        raise ValueError("Input contains NaN or infinity.")
    return _set_measure(A != 0, 1, ctx)


def hamming_similarity(feature_matrix, ctx=None) -> np.ndarray:
    """What the reference's metric="jaccard" branch really computes (snekmer/score.py:166-168):
    ``1 - pairwise_distances(X, metric="hamming")`` = fraction of columns on which two rows are EQUAL, for any
    matrix: the k-mer count matrix the docstring names, the binary ``vecs``, or real-valued features.  float64,
    the same arithmetic as scipy's hamming (count / columns in double)."""
    return _set_measure(feature_matrix, 0, ctx)


def _pattern_and_value_csr(ctx, A):
    """(0/1 pattern CSR, CSR over distinct (column, value) pairs or None when A is 0/1, non-zeros per row)."""
    rows, cols = np.nonzero(A)
    vals = A[rows, cols]
    n = A.shape[0]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])

    def csr(colids):
        nnz = int(colids.size)
        c = engine.CountsCSR(ctx, n, nnz, 32, ctx.to_device(indptr), ctx.to_device(np.zeros(max(nnz, 1), dtype=np.uint32)),
                             ctx.to_device(np.ones(max(nnz, 1), dtype=np.uint32)), None)
        c.colidx = ctx.to_device(colids.astype(np.uint32) if nnz else np.zeros(1, np.uint32))
        return c

    pattern = csr(cols)
    binary = A.dtype == bool or not vals.size or bool(np.all(vals == 1))
    if binary:
        return pattern, None, 0, np.diff(indptr).astype(np.uint32)
    # one column per distinct (column, value) pair: two rows share such a column iff they hold the SAME non-zero value
    _, vrank = np.unique(vals, return_inverse=True)
    key = cols.astype(np.int64) * (int(vrank.max()) + 1) + vrank
    uniq, pair_col = np.unique(key, return_inverse=True)
    if uniq.size >= 2**32 - 1:
        raise OverflowError("2^32 distinct (column, value) pairs or more")
    return pattern, csr(pair_col), int(uniq.size), np.diff(indptr).astype(np.uint32)


def _set_measure(feature_matrix, kind: int, ctx=None) -> np.ndarray:
    import ctypes as C

    from . import _hip

    ctx = ctx or _hip.default_context()
    A = _plain(feature_matrix)
    if hasattr(A, "toarray"):
        A = A.toarray()
    A = np.asarray(A)
    if A.ndim != 2:
        raise ValueError("expected a 2-D feature matrix")
    if A.dtype.kind not in "buif":
        A = A.astype(np.float64)  # sklearn / scipy convert to double too; raises for non-numeric input
    if A.dtype.kind == "f" and not np.all(np.isfinite(A)):
        raise ValueError("Input contains NaN or infinity.")
    n, ncols = A.shape
    if ncols == 0:
        raise ValueError("feature matrix has no columns")
    if n == 0:
        return np.zeros((0, 0), dtype=np.float64)
    pattern, valued, nvalcols, nnz_row = _pattern_and_value_csr(ctx, A)
    ones = ctx.to_device(np.ones(n + 4, dtype=np.float32))
    ld = (n + 3) // 4 * 4

    keep = []  # every array a queued kernel reads stays referenced until the result has been copied back below

    def gram(x, width):  # exact integer Gram in float32 cells (< 2^24: checked by skm_setsim_f64)
        colptr, post = engine.transpose(ctx, n, x.nnz, width, x.rowptr, x.colidx, x.counts)
        keep.extend((colptr, post))
        return engine.cosine_matrix(ctx, x, ones, n, width, colptr, post, ones, mode=0, ld=ld)

    both = gram(pattern, ncols)
    equal = gram(valued, nvalcols) if valued is not None else None
    d_nnz = ctx.to_device(np.concatenate([nnz_row, np.zeros(4, dtype=np.uint32)]))
    out = ctx.empty((n, n), np.float64)
    ctx.call("skm_setsim_f64", kind, C.c_int64(n), C.c_int64(n), C.c_int64(ncols), C.c_void_p(d_nnz.ptr), C.c_void_p(d_nnz.ptr),
             C.c_void_p(both.ptr), C.c_void_p(equal.ptr if equal is not None else None), C.c_int64(ld), C.c_void_p(out.ptr),
             C.c_int64(n))
    result = out.download().reshape(n, n)  # waits for the stream: everything in `keep` has been read
    del keep[:]
    return result
