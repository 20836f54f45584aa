#!/usr/bin/env python3
"""Randomised differential check of the HIP path against the oracle (test infrastructure; run by hand or by
tests/test_gpu_parity.py::test_fuzz_short on a GPU box):  python tests/fuzz_parity.py [seconds] [first_seed]

Every round draws an alphabet, a k, and a batch (family sequences with substitutions, junk characters, trailing
stars, lowercase, empties, repeats, a few long records, now and then a homopolymer long enough for dot products
beyond int32) and compares, bit for bit unless stated:
  * skm_count_csr and skm_vectorize_csr (CSR, basis codes, column starts, column ids, row norms) with the C oracle;
  * the N x N cosine by the neighbour-list path, by the cursor kernel and by the overlapped schedule with each other,
    and with the oracle's float64 rows to 1e-5; every eighth round also as a stream of two batches through
    engine.OverlappedPipeline with part of the neighbour lists built on its side contexts;
  * neighbour lists (skm_gram_neighbors) of a random row block: neighbour sets and exact integer dot products;
  * every fourth round: the rule body (`vectorize_records`: first-seen basis, min_filter, presence rows, reduced
    strings, explicit basis) against the oracle's restatement of rules/kmerize.smk:67-139;
  * every fourth round: the sklearn call sites (cosine_similarity on ndarrays / DataFrames / scipy CSR, real-valued
    matrices, connection_matrix_from_features, Jaccard) against scikit-learn and scipy themselves;
  * every eighth round: the per-record Python surface (reduce, reduce_vectorize, make_feature_matrix,
    KmerBasis.transform) against the oracle's restatements of snekmer/vectorize.py, string for string;
  * every fourth round: the learn/apply chain (group sums, fused top-2 epilogue) against float64 numpy;
  * every fourth round: the dense int8 matrix-core cosine at a random shape against the integer Gram (numpy)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
JUNK = np.frombuffer(b"XBZUO*-acgt.", dtype=np.uint8)


def draw_batch(rng, big=False):
    n_fam = int(rng.integers(20, 60)) if big else int(rng.integers(1, 12))
    seqs = []
    for _ in range(n_fam):
        L = int(rng.choice([rng.integers(0, 40), rng.integers(40, 700), rng.integers(700, 3000)], p=[0.1, 0.8, 0.1]))
        if rng.random() < 0.03:
            L = int(rng.integers(5000, 22000))
        low = rng.random() < 0.15  # low-complexity family: few distinct residues, many repeated k-mers
        root = AA[rng.integers(0, 3 if low else 20, size=L)]
        members = int(rng.integers(20, 90)) if big else int(rng.integers(1, 40))
        if big and len(seqs) < 3000 and rng.random() < 0.04:
            # one family of thousands: every member has more neighbours than the first sparse pass holds (1536), so
            # the rows go through k_cosine_heavy, and skm_gram_neighbors through its large-table tiers
            members, L = int(rng.integers(1600, 2600)), int(rng.integers(40, 200))
            root = AA[rng.integers(0, 20, size=L)]
        for _ in range(members):
            s = root.copy()
            if L:
                m = rng.random(L) < rng.choice([0.0, 0.02, 0.1, 0.3])
                s[m] = AA[rng.integers(0, 20, size=int(m.sum()))]
                if rng.random() < 0.2:
                    j = rng.random(L) < 0.01
                    s[j] = JUNK[rng.integers(0, len(JUNK), size=int(j.sum()))]
            t = s.tobytes().decode()
            if rng.random() < 0.1:
                t += "*" * int(rng.integers(1, 3))
            seqs.append(t)
    if rng.random() < 0.08:
        # records of 46 342 windows or more in one or two letters: a k-mer count whose square leaves int32, so the row's
        # dot products take the float64-accumulator kernel (and its strip neighbours with it)
        for _ in range(int(rng.integers(1, 3))):
            L = int(rng.integers(46400, 70000))
            unit = AA[rng.integers(0, 20, size=int(rng.integers(1, 3)))]
            s = np.resize(unit, L).copy()
            if rng.random() < 0.5:
                m = rng.random(L) < 0.0005
                s[m] = AA[rng.integers(0, 20, size=int(m.sum()))]
            seqs.append(s.tobytes().decode() + ("*" if rng.random() < 0.3 else ""))
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


def one_round(ctx, seed, verbose=False):
    from oracle import c_oracle as orc
    from snekmer_amd import _hip
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.utils import pack_sequences

    rng = np.random.default_rng(seed)
    names = ["hydro", "standard", "solvacc", "hydrocharge", "hydrostruct", "miqs", "ptm", "None", "red6"]
    name = names[int(rng.integers(0, len(names)))]
    lut = A.build_lut(None if name == "None" else name)
    kmax = 1
    while lut.nsym ** (kmax + 1) < 2**64 and kmax < 32:
        kmax += 1
    k = int(rng.integers(1, kmax + 1))
    seqs = draw_batch(rng, big=seed % 16 == 7)  # every 16th round: thousands of rows (lists by default, overflow passes)
    res, off = pack_sequences(seqs)
    n = len(seqs)
    tag = f"seed {seed}: {name} k={k} n={n} residues={len(res)}"
    if verbose:
        print(tag, flush=True)
    batch = engine.SeqBatch(ctx, res, off)
    o_rowptr, o_codes, o_counts, o_first = orc.count_csr(lut.rank, lut.nsym, k, res, off)
    ob, odf, otot, ofk, ocol = orc.basis(o_rowptr, o_codes, o_counts, o_first)
    nnz = len(o_codes)

    def same(a, b, what):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and (a == b).all(), f"{tag}: {what} differs"

    pipes = {"three": engine.Pipeline(ctx, lut, k, fused=False, dense_route=False)}  # always the sparse kernels
    if len(res) >= 1:
        pipes["fused"] = engine.Pipeline(ctx, lut, k, fused=True)
    S = {}
    for key, p in pipes.items():
        out = p.step(batch)
        S[key] = out.download().reshape(out.shape)[:n, :n].copy()
        same(p.csr.rowptr.download(n + 1), o_rowptr, f"{key} rowptr")
        assert p.csr.nnz == nnz, f"{tag}: {key} nnz {p.csr.nnz} != {nnz}"
        same(p.csr.codes.download(nnz).astype(np.uint64), o_codes, f"{key} codes")
        same(p.csr.counts.download(nnz), o_counts, f"{key} counts")
        assert p.basis.ncols == len(ob), f"{tag}: {key} ncols"
        same(p.basis.codes.download(len(ob)).astype(np.uint64), ob, f"{key} basis")
        cp = p.basis.colptr.download(len(ob) + 1).astype(np.int64)
        same(np.diff(cp), odf, f"{key} df")
        col = p.csr.colidx.download(nnz)
        keep = col != 0xFFFFFFFF  # k-mers of one row only carry no column id in the cosine pipeline
        same(col[keep], ocol[keep], f"{key} colidx")
        assert (odf[ocol[~keep]] == 1).all(), f"{tag}: {key} elided a shared column"
    if "fused" in S:
        if pipes["fused"].route == "dense":  # small full basis: int8 GEMM on the matrix cores (+ exact fix-up rows);
            # mirrored tiles multiply the two norms in the other order: one float32 rounding
            err = float(np.abs(S["fused"] - S["three"]).max()) if n else 0.0
            assert err <= 1e-6, f"{tag}: dense route vs sparse route {err}"
        else:
            same(S["fused"], S["three"], "fused vs three-call cosine")
    p = pipes["three"]
    if n and nnz:
        rows = np.arange(n) if n <= 400 else np.sort(rng.choice(n, 400, replace=False))
        ref = orc.cosine_rows(o_rowptr, ocol, o_counts, len(ob), rows)
        err = float(np.abs(S["three"][rows] - ref).max())
        assert err <= 1e-5, f"{tag}: cosine vs oracle {err}"
        for env in ({"SKM_COSINE_PATH": "cursor"}, {"SKM_COSINE_PATH": "lists"}, {"SKM_COSINE_OVERLAP": "1", "SKM_COSINE_PATH": "lists"},
                    {"SKM_COSINE_PATH": "lists", "SKM_HEAVY_PACK": "1"}, {"SKM_COSINE_PATH": "lists", "SKM_HEAVY_PACK": "1", "SKM_HEAVY_PANEL": "1"}):
            with _hip.options(**env):
                alt = p.cosine().download().reshape(p.out.shape)[:n, :n]
            same(alt, S["three"], f"cosine under {env}")
        if seed % 8 == 5:
            # a stream of two batches through OverlappedPipeline with the lists of a random share of the rows built on the
            # side contexts (skm_cosine_csr_phase), forced on whatever the size: the one-stream result both times
            op = engine.OverlappedPipeline(ctx, lut, k, side_list_fraction=float(rng.choice([0.25, 0.6, 1.0])))
            op.SPLIT_MIN_ROWS = 1
            op.prefetch(batch)
            for nxt in (batch, None):
                got = op.step(nxt)
                op.sync()
                same(got.download().reshape(got.shape)[:n, :n], S["three"], "OverlappedPipeline with split lists")
            op.close()
        # neighbour lists of a random row block: the same neighbour set and the exact integer dot products
        lo = int(rng.integers(0, n))
        hi = int(rng.integers(lo + 1, n + 1))
        b = p.basis
        nb = engine.gram_neighbors(ctx, p.csr, p.rnorm, n, b.ncols, b.colptr, b.post, p.rnorm, row0=lo, row1=hi, post_bits=b.post_bits, postcnt=b.postcnt)
        start, length, jj, dot = nb.host()
        nsq = np.add.reduceat(np.concatenate([o_counts.astype(np.float64) ** 2, [0.0]]), np.minimum(o_rowptr[:-1], nnz))
        nsq[np.diff(o_rowptr) == 0] = 0.0
        norms = np.sqrt(np.where(nsq > 0, nsq, 1.0))
        full = orc.cosine_rows(o_rowptr, ocol, o_counts, len(ob), np.arange(lo, hi))
        for r in range(hi - lo):
            if length[r] == 0xFFFFFFFF:
                continue  # a row the list kernels cannot hold (the cosine path computes it with the cursor kernel)
            js = jj[int(start[r]) : int(start[r]) + int(length[r])]
            ds = dot[int(start[r]) : int(start[r]) + int(length[r])]
            assert sorted(js.tolist()) == np.nonzero(full[r] > 0)[0].tolist(), f"{tag}: neighbour set of row {lo + r}"
            if len(js):
                gram = full[r, js] * norms[lo + r] * norms[js]
                assert np.abs(ds - np.rint(gram)).max() == 0, f"{tag}: dot products of row {lo + r}"
    return tag


def apply_round(ctx, seed):
    """Learn / apply chain on a random batch: per-annotation sums of count rows (skm_csr_group_sum) against a dense
    numpy sum, and the fused top-2 epilogue (skm_apply_top2) against float64 numpy on the dense matrices: indices
    (score desc, column asc), exact integer dots, scores to 1e-12."""
    from oracle import c_oracle as orc
    from snekmer_amd import alphabet as A
    from snekmer_amd import apply as skm_apply
    from snekmer_amd import engine
    from snekmer_amd.utils import pack_sequences

    rng = np.random.default_rng(seed)
    name = ["hydro", "standard", "solvacc", "miqs", "red6"][int(rng.integers(0, 5))]
    lut = A.build_lut(name)
    k = int(rng.integers(2, 9))
    seqs = [s for s in draw_batch(rng) if len(s) < 3000]
    if len(seqs) < 2:
        return f"apply seed {seed}: skipped"
    res, off = pack_sequences(seqs)
    n = len(seqs)
    batch = engine.SeqBatch(ctx, res, off)
    csr = engine.count_csr(ctx, batch, lut, k)
    with_post = seed % 8 == 2  # every other apply round: the count matrix's own postings feed the aggregation (no sort)
    basis = engine.build_basis(ctx, csr, lut.nsym, k, postings=with_post)
    B = basis.ncols
    tag = f"apply seed {seed}: {name} k={k} n={n} B={B}" + (" postings" if with_post else "")
    if B == 0 or B * n > 3e7:
        return tag + " skipped"
    o_rowptr, o_codes, o_counts, o_first = orc.count_csr(lut.rank, lut.nsym, k, res, off)
    ob, odf, otot, ofk, ocol = orc.basis(o_rowptr, o_codes, o_counts, o_first)
    X = np.zeros((n, B), dtype=np.int64)
    rows = np.repeat(np.arange(n), np.diff(o_rowptr))
    X[rows, ocol] = o_counts
    ng = int(rng.integers(1, min(n, 40) + 1))
    groups = rng.integers(0, ng, size=n)
    T = np.zeros((ng, B), dtype=np.int64)
    np.add.at(T, groups, X)
    totals = skm_apply.group_sum(ctx, csr, groups, ng, basis=basis if with_post else None, ncols=B)
    rp = totals.rowptr.download(ng + 1)
    tc, tv = totals.colidx.download(totals.nnz), totals.counts.download(totals.nnz)
    Td = np.zeros((ng, B), dtype=np.int64)
    Td[np.repeat(np.arange(ng), np.diff(rp)), tc] = tv
    assert (Td == T).all(), f"{tag}: group sums"
    idx, score, dot = skm_apply.apply_top2(ctx, csr, B, totals)
    G = X @ T.T
    nx, ny = np.sqrt((X * X).sum(axis=1).astype(np.float64)), np.sqrt((T * T).sum(axis=1).astype(np.float64))
    nx[nx == 0] = 1.0
    ny[ny == 0] = 1.0
    S = G / (nx[:, None] * ny[None, :])
    for r in range(n):
        order = np.lexsort((np.arange(ng), -S[r]))[:2]
        for slot in range(min(2, ng)):
            j = int(idx[r, slot])
            assert j < ng, f"{tag}: row {r} slot {slot} has no column"
            # equal scores may be told apart differently in the last bit: compare values, then ids where clear
            assert abs(score[r, slot] - S[r, order[slot]]) <= 1e-12, f"{tag}: row {r} slot {slot} score"
            assert dot[r, slot] == G[r, j], f"{tag}: row {r} slot {slot} dot"
            assert abs(S[r, j] - score[r, slot]) <= 1e-12, f"{tag}: row {r} slot {slot} column {j}"
    return tag


def records_round(ctx, seed):
    """The rule body (rules/kmerize.smk:67-139) on random records: snekmer_amd.kmerize.vectorize_records against the
    oracle's restatement of the rule (oracle/ref_path.kmerize_rule: first-seen basis, `> min_filter`, np.isin
    presence rows, reduced strings, ids, raw lengths), with and without an explicit basis."""
    from oracle import ref_path as R
    from snekmer_amd import alphabet as A
    from snekmer_amd.kmerize import vectorize_records

    rng = np.random.default_rng(seed)
    name = ["hydro", "standard", "solvacc", "hydrocharge", "hydrostruct", "miqs", "ptm", "None"][int(rng.integers(0, 8))]
    key = None if name == "None" else name
    table = A.FULL_ALPHABETS[name]
    k = int(rng.integers(1, 7))
    seqs = [s[:400] for s in draw_batch(rng)][:60]
    records = [(f"rec{i}|x", s) for i, s in enumerate(seqs)]
    min_filter = int(rng.integers(0, 3))
    tag = f"records seed {seed}: {name} k={k} n={len(records)} min_filter={min_filter}"
    ref = R.kmerize_rule(records, k, table, min_filter=min_filter)
    got = vectorize_records(records, key, k, min_filter=min_filter, ctx=ctx)
    for f in ("kmerlist", "ids", "seqs", "lengths"):
        assert [str(x) for x in got[f]] == [str(x) for x in ref[f]], f"{tag}: {f}"
    assert got["vecs"].shape == ref["vecs"].shape and (got["vecs"] == ref["vecs"]).all(), f"{tag}: vecs"
    if len(ref["kmerlist"]) >= 2:
        # the basis.txt branch: an explicit basis (shuffled subset plus a k-mer nobody has), min_filter ignored
        kl = [str(x) for x in ref["kmerlist"]]
        basis = [kl[i] for i in rng.permutation(len(kl))[: max(1, len(kl) // 2)]]
        ref2 = R.kmerize_rule(records, k, table, min_filter=min_filter, basis=basis)
        got2 = vectorize_records(records, key, k, min_filter=min_filter, basis=basis, ctx=ctx)
        assert [str(x) for x in got2["kmerlist"]] == basis, f"{tag}: explicit basis order"
        assert (got2["vecs"] == ref2["vecs"]).all(), f"{tag}: vecs with an explicit basis"
    return tag


def score_round(ctx, seed):
    """The sklearn call sites (rules/apply.smk:282, score.py:166-171) against scikit-learn itself on random matrices:
    cosine_similarity for count matrices (ndarray, DataFrame, scipy CSR; every kernel path) to 1e-5 and for
    real-valued matrices to 1e-12, connection_matrix_from_features for "cosine" and the "jaccard" (= 1 - hamming)
    branch, jaccard_distance against scipy."""
    import pandas as pd
    import scipy.sparse as sp
    from scipy.spatial.distance import pdist, squareform
    from sklearn.metrics import pairwise_distances
    from sklearn.metrics.pairwise import cosine_similarity as sk_cos

    import snekmer_amd as skm

    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    K = int(rng.choice([1, 7, 64, 200, 1024, 4096]))
    dens = float(rng.choice([0.01, 0.1, 0.6]))
    X = (rng.random((n, K)) < dens) * rng.integers(1, int(rng.choice([2, 5, 127, 300])), size=(n, K))
    Y = (rng.random((m, K)) < dens) * rng.integers(1, 5, size=(m, K))
    X[rng.integers(0, n)] = 0  # a zero row: similarity 0 with everything, itself included
    tag = f"score seed {seed}: n={n} m={m} K={K} density={dens}"
    ref = sk_cos(X, Y)
    for path in ("auto", "sparse"):
        got = skm.score.cosine_similarity(X, Y, ctx=ctx, path=path)
        assert got.dtype == np.float64 and np.abs(got - ref).max() <= 1e-5, f"{tag}: counts, path {path}"
    got = skm.score.cosine_similarity(pd.DataFrame(X), pd.DataFrame(Y), ctx=ctx)
    assert np.abs(got - ref).max() <= 1e-5, f"{tag}: DataFrames"
    got = skm.score.cosine_similarity(sp.csr_matrix(X), sp.csr_matrix(Y), ctx=ctx)
    assert np.abs(got - ref).max() <= 1e-5, f"{tag}: scipy CSR"
    got = skm.score.cosine_similarity(X, ctx=ctx)
    assert np.abs(got - sk_cos(X)).max() <= 1e-5, f"{tag}: X with itself"
    Xf = X / np.maximum(X.sum(axis=1, keepdims=True), 1) + rng.random((n, K)) * (rng.random() < 0.5)
    got = skm.score.cosine_similarity(Xf, Y.astype(np.float64) * 0.5, ctx=ctx)
    # (a draw in which both happen to hold integers only is a COUNT matrix to the product: exact integer dots scaled
    # in float32, the 1e-5 contract; seed 122947 is one)
    both_integral = bool(np.all(Xf == np.floor(Xf)) and np.all(Y * 0.5 == np.floor(Y * 0.5)))
    assert np.abs(got - sk_cos(Xf, Y * 0.5)).max() <= (1e-5 if both_integral else 1e-12), f"{tag}: real-valued"
    D = skm.score.connection_matrix_from_features(X, metric="cosine")
    assert np.abs(D - pairwise_distances(X, metric="cosine")).max() <= 1e-5 and (np.diag(D) == 0).all(), f"{tag}: cosine distance"
    Df = skm.score.connection_matrix_from_features(Xf, metric="cosine")
    assert np.abs(Df - pairwise_distances(Xf, metric="cosine")).max() <= 1e-12, f"{tag}: cosine distance, real-valued"
    import warnings

    from snekmer_amd.score import PAIRWISE_METRICS

    metric = sorted(PAIRWISE_METRICS)[int(rng.integers(0, len(PAIRWISE_METRICS)))]
    Xm = Xf if rng.random() < 0.5 else X.astype(np.float64)
    if metric == "haversine":  # two columns of radians
        Xm = np.stack([rng.uniform(-np.pi / 2, np.pi / 2, len(Xm)), rng.uniform(-np.pi, np.pi, len(Xm))], axis=1)
    elif metric == "nan_euclidean":  # missing values, now and then a row without any
        Xm = Xm.copy()
        Xm[rng.random(Xm.shape) < 0.25] = np.nan
        if len(Xm) and rng.random() < 0.3:
            Xm[int(rng.integers(0, len(Xm)))] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = pairwise_distances(Xm, metric=metric)
    got = skm.score.connection_matrix_from_features(Xm, metric=metric)
    assert (np.isnan(got) == np.isnan(want)).all(), f"{tag}: {metric} nan pattern"
    scale = max(1.0, float(np.nanmax(np.abs(want), initial=0.0)))
    if metric == "nan_euclidean":
        # sklearn expands (x - y)^2 = xx + yy - 2 xy here too (with the missing columns zeroed), then scales by K / present:
        # the same cancellation error, absolute in the squared distance, times that scale (at most K)
        Z = np.nan_to_num(Xm)
        bound = 1e-12 * max(1.0, 2.0 * float((Z * Z).sum(axis=1).max(initial=0.0))) * Xm.shape[1]
        assert np.nanmax(np.abs(got * got - want * want), initial=0.0) <= bound, f"{tag}: {metric}"
    elif metric in ("euclidean", "l2"):
        # sklearn takes sqrt(xx + yy - 2 xy), which cancels for close rows; the kernel sums (x - y)^2 directly.  The two
        # agree to the expansion's own rounding error, which is absolute in the SQUARED distance: eps * (xx + yy)
        assert np.nanmax(np.abs(got * got - want * want), initial=0.0) <= 1e-12 * max(1.0, 2.0 * float((Xm * Xm).sum(axis=1).max())), \
            f"{tag}: {metric}"
    else:
        assert np.nanmax(np.abs(got - want), initial=0.0) <= 1e-11 * scale, f"{tag}: {metric}"
    Bm = X > 0
    for M, what in ((Bm, "binary"), (X, "counts"), (np.round(Xf, 1), "real-valued")):
        H = skm.score.connection_matrix_from_features(M)  # metric="jaccard": 1 - hamming upstream
        assert (H == 1 - pairwise_distances(M, metric="hamming")).all(), f"{tag}: 1 - hamming, {what}"
        if n >= 2:
            J = skm.score.jaccard_distance(M)
            assert (J == squareform(pdist(M, "jaccard"))).all(), f"{tag}: jaccard distance, {what}"
    return tag


def surface_round(ctx, seed):
    """The per-record Python surface an unchanged rule body calls (vectorize.reduce, KmerVec.reduce_vectorize and its
    batch form, make_feature_matrix, KmerBasis.transform / harmonize) against the oracle's restatements of
    snekmer/vectorize.py, string for string."""
    from oracle import ref_path as R

    import snekmer_amd as skm
    from snekmer_amd import alphabet as A

    rng = np.random.default_rng(seed)
    name = ["hydro", "standard", "solvacc", "hydrocharge", "hydrostruct", "miqs", "ptm", "None"][int(rng.integers(0, 8))]
    key = None if name == "None" else name
    table = A.FULL_ALPHABETS[name]
    k = int(rng.integers(1, 9))
    seqs = [s[:300] for s in draw_batch(rng)][:25] + ["", "M", "*", "MKV*", "mkvl"]
    tag = f"surface seed {seed}: {name} k={k} n={len(seqs)}"
    kv = skm.vectorize.KmerVec(alphabet=key, k=k)
    vecs_ref = []
    for sq in seqs:
        assert skm.vectorize.reduce(sq, alphabet=key, mapping=A.FULL_ALPHABETS) == R.reduce(sq, table), f"{tag}: reduce"
        got, ref = kv.reduce_vectorize(sq), R.reduce_vectorize(sq, k, table)
        assert got.shape == ref.shape and [str(x) for x in got] == [str(x) for x in ref], f"{tag}: reduce_vectorize"
        vecs_ref.append(ref)
    for got, ref in zip(kv.reduce_vectorize_batch(seqs), vecs_ref):
        assert [str(x) for x in got] == [str(x) for x in ref], f"{tag}: reduce_vectorize_batch"
    if sum(len(v) for v in vecs_ref):
        mf = int(rng.integers(0, 3))
        rows, kl = skm.vectorize.make_feature_matrix(vecs_ref, mf)
        rrows, rkl = R.make_feature_matrix(vecs_ref, mf)
        assert [str(x) for x in kl] == [str(x) for x in rkl], f"{tag}: make_feature_matrix kmerlist"
        assert len(rows) == len(rrows) and all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(rows, rrows)), f"{tag}: rows"
        if len(rkl) >= 2:
            # KmerBasis.transform: columns from one k-mer order into another, unknown k-mers -> zero columns
            vb = [str(x) for x in rkl]
            mat = rng.integers(0, 5, size=(4, len(vb))).astype(np.float64)
            target = [vb[i] for i in rng.permutation(len(vb))[: max(1, len(vb) - 1)]] + ["?" * k]
            kb = skm.vectorize.KmerBasis()
            kb.set_basis(target)
            assert (kb.transform(mat, vb) == R.basis_transform(target, mat, vb)).all(), f"{tag}: KmerBasis.transform"
    return tag


def dense_round(ctx, seed):
    """Dense int8 cosine on the matrix cores: random shapes around the kernels' switch points (register-staged,
    128 x 128, 256 x 256 staggered; rectangular, X is Y), exact integer Gram with unit norms."""
    from snekmer_amd import engine

    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(1, 300), rng.integers(900, 1200), rng.integers(1024, 2600)]))
    m = int(rng.choice([rng.integers(1, 300), rng.integers(900, 1200), rng.integers(1024, 2600)]))
    kdim = 64 * int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 20]))
    sym = rng.random() < 0.35
    if sym:
        m = n
    X = rng.integers(-8, 8, size=(n, kdim)).astype(np.int8)
    Y = X if sym else rng.integers(-8, 8, size=(m, kdim)).astype(np.int8)
    X[rng.integers(0, n)] = 0
    ld = int(rng.choice([(m + 3) // 4 * 4, m, m + 1, m + 7]))
    ones = ctx.to_device(np.ones(max(n, m) + 4, dtype=np.float32))
    dX = ctx.to_device(X)
    dY = dX if sym else ctx.to_device(Y)
    out = engine.cosine_dense_i8(ctx, n, m, kdim, dX, dY, ones, ones, ld=ld)
    G = out.download().reshape(-1, ld)[:n, :m]
    ref = X.astype(np.int64) @ Y.astype(np.int64).T
    assert (G.astype(np.int64) == ref).all(), f"dense seed {seed}: n={n} m={m} k={kdim} ld={ld} sym={sym}"
    return f"dense seed {seed}: n={n} m={m} k={kdim} ld={ld} sym={sym}"


def main():
    from snekmer_amd import _hip
    from snekmer_amd import alphabet as A

    if "red6" not in A.ALPHABETS:
        A.register_alphabet("red6", A.RED6_GROUPS)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = _hip.default_context()
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget:
        one_round(ctx, seed, verbose=True)
        if seed % 4 == 0:
            print(dense_round(ctx, seed), flush=True)
        if seed % 4 == 2:
            print(apply_round(ctx, seed), flush=True)
        if seed % 4 == 1:
            print(records_round(ctx, seed), flush=True)
        if seed % 4 == 3:
            print(score_round(ctx, seed), flush=True)
        if seed % 8 == 5:
            print(surface_round(ctx, seed), flush=True)
        seed += 1
        done += 1
    print(f"fuzz ok: {done} rounds in {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
