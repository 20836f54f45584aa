"""kmerize: the body of the reference's ``rule vectorize`` (snekmer/rules/kmerize.smk:67-142)
as one function over one FASTA file, with the per-record Python loops replaced by batched
device calls.

Outputs follow the rule's ``.npz`` contract: ``kmerlist`` (observed k-mers in
first-seen order, kept iff total occurrences > min_filter; or the given basis verbatim),
``vecs`` (float64 0/1 presence), ``seqs`` (reduced strings), ``ids``, ``lengths`` (raw length,
trailing '*' included) -- plus, additively, the integer count matrix in CSR form, which the
reference recomputes later in Python (rules/learn.smk:359-383).

Known deviation: with an explicit `basis` holding strings that are not k class letters, the
reference's *count* loop would still count such substrings of the reduced string; here (and in
the reference's own presence matrix) they never match.
"""
import pickle
import time
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import engine
from .alphabet import FULL_ALPHABETS, build_lut
from .io import dump_kmers, read_fasta, read_fasta_packed, save_npz, save_npz_sparse
from .utils import pack_sequences
from .vectorize import KmerVec, _restore_wide_chars


def vectorize_records(
    records: Sequence[Tuple[str, str]],
    alphabet: Union[str, int, None],
    k: int,
    min_filter: int = 0,
    basis: Optional[Sequence[str]] = None,
    dense: bool = True,
    ctx=None,
    timings: Optional[dict] = None,
) -> Dict[str, np.ndarray]:
    """The rule body over (id, sequence) pairs.  Strings are packed once; everything after that is
    `vectorize_packed`.  Characters outside latin-1 (carried through `reduce` by position) are put back
    afterwards."""
    t0 = time.perf_counter()
    ids = [r[0] for r in records]
    raw = [str(r[1]) for r in records]
    res, off = pack_sequences(raw)
    if timings is not None:
        timings["pack_s"] = timings.get("pack_s", 0.0) + time.perf_counter() - t0
    out = vectorize_packed(np.asarray(ids, dtype=str) if ids else np.array([], dtype=str), res, off, alphabet, k,
                           min_filter=min_filter, basis=basis, dense=dense, ctx=ctx, timings=timings)
    wide = [i for i, s in enumerate(raw) if not s.isascii() and any(ord(c) > 255 for c in s)]
    if wide:
        seqs = out["seqs"].tolist()
        for i in wide:
            seqs[i] = _restore_wide_chars(raw[i], seqs[i])
        out["seqs"] = np.asarray(seqs, dtype=str)
    return out


def vectorize_packed(
    ids: np.ndarray,
    residues: np.ndarray,
    offsets: np.ndarray,
    alphabet: Union[str, int, None],
    k: int,
    min_filter: int = 0,
    basis: Optional[Sequence[str]] = None,
    dense: bool = True,
    ctx=None,
    timings: Optional[dict] = None,
) -> Dict[str, np.ndarray]:
    """The rule body (snekmer/rules/kmerize.smk:67-139) over a packed batch (`ids` '<U' array, residues uint8, offsets
    int64[n+1]: what io.read_fasta_packed returns).  No per-record and no per-k-mer Python: reduced strings, the
    k-mer strings of the basis and the count matrix in kmerlist order all come off the device as arrays.
    `timings` (optional dict) receives gpu_s / decode_s, the seconds spent in device calls that produce integers
    and in the calls that only format them as strings."""
    from . import _hip

    ctx = ctx or _hip.default_context()
    lut = build_lut(alphabet)
    t0 = time.perf_counter()
    batch = engine.SeqBatch(ctx, residues, offsets)
    n = batch.n

    # counts + observed basis
    csr = engine.count_csr(ctx, batch, lut, k, with_firstpos=True)
    b = engine.build_basis(ctx, csr, lut.nsym, k, stats=True, first_seen=True, postings=False)
    B = b.ncols
    d_keep = None
    if basis is None:
        # columns in first-seen order, kept iff total occurrences > min_filter (kmerize.smk:102-104): on the device
        import ctypes as C

        d_keep = ctx.empty(max(B, 1), np.uint32)
        d_colmap = ctx.empty(max(B, 1), np.uint32)
        nkeep = C.c_int64(0)
        ctx.call("skm_basis_select", C.c_int64(B), C.c_void_p(b.fs_order.ptr), C.c_void_p(b.total.ptr), C.c_uint64(max(int(min_filter), 0)),
                 C.c_void_p(d_keep.ptr), C.c_void_p(d_colmap.ptr), C.byref(nkeep))
        ncols_out = int(nkeep.value)  # (a negative min_filter keeps everything, like 0: every total is >= 1)
        kmerlist = None
    else:
        bcodes = b.codes.download(B).astype(np.uint64)
        kmerlist = np.asarray(list(basis))
        want, ok = lut.encode([str(x) for x in kmerlist], k)
        pos = np.searchsorted(bcodes, want)
        pos_c = np.clip(pos, 0, max(B - 1, 0))
        hit = ok & (pos < B) & (bcodes[pos_c] == want if B else False)
        colmap = np.full(max(B, 1), 0xFFFFFFFF, dtype=np.uint32)
        # A k-mer listed more than once: upstream every one of its columns is set (np.isin over the basis,
        # kmerize.smk:119; the count loop of learn.smk:376-383 looks each column up by name).  The device fills the
        # FIRST of them; the others are copies made below.
        _, first_of = np.unique(kmerlist, return_index=True) if len(kmerlist) else (None, np.zeros(0, np.int64))
        lead = np.zeros(len(kmerlist), dtype=bool)
        lead[first_of] = True
        colmap[pos_c[hit & lead]] = np.nonzero(hit & lead)[0].astype(np.uint32)
        ncols_out = int(len(kmerlist))
        d_colmap = ctx.to_device(colmap)
    # additive: integer counts in the kmerlist column order, CSR
    c_rowptr, c_col, c_val = engine.csr_remap_columns(ctx, csr, d_colmap, B)
    vecs = None
    if dense:
        vecs = engine.csr_to_dense(ctx, n, csr.rowptr, csr.colidx, csr.counts, ncols_out, colmap=d_colmap,
                                   presence=True, dtype=np.float64).download()
        vecs = vecs.reshape(max(n, 1), max(ncols_out, 1))[:n, :ncols_out]
    if basis is not None and not lead.all():
        c_rowptr, c_col, c_val, vecs = _copy_repeated_columns(kmerlist, c_rowptr, c_col, c_val, vecs)
    t1 = time.perf_counter()

    # string forms: reduced sequences (kmerize.smk:121-127) and the basis k-mers (kmerize.smk:102-106)
    seqs = engine.reduced_strings(ctx, batch, lut)
    if kmerlist is None:
        kmerlist = engine.decode_kmers(ctx, lut, k, b.codes, ncols_out, d_keep)
    t2 = time.perf_counter()
    if timings is not None:
        timings["gpu_s"] = timings.get("gpu_s", 0.0) + t1 - t0
        timings["decode_s"] = timings.get("decode_s", 0.0) + t2 - t1

    out = {
        "kmerlist": kmerlist if len(kmerlist) else np.array([], dtype=str),
        "ids": np.asarray(ids, dtype=str) if n else np.array([], dtype=str),
        "seqs": seqs,
        "lengths": np.diff(np.asarray(offsets, dtype=np.int64)),
    }
    if vecs is not None:
        out["vecs"] = vecs
    out["counts_rowptr"], out["counts_col"], out["counts_val"] = c_rowptr, c_col, c_val
    return out


def _copy_repeated_columns(kmerlist, rowptr, col, val, vecs):
    """Columns of an explicit basis that repeat an earlier k-mer become copies of that k-mer's first column: in the
    presence matrix and in the CSR counts (each entry of a first column is followed by its copies, columns ascending)."""
    _, inverse = np.unique(kmerlist, return_inverse=True)
    order = np.argsort(inverse, kind="stable")          # columns grouped by k-mer, ascending within a group
    group_start = np.zeros(int(inverse.max()) + 2, dtype=np.int64)
    np.cumsum(np.bincount(inverse), out=group_start[1:])
    first = order[group_start[inverse]]                 # first column of every column's k-mer
    if vecs is not None:
        vecs = vecs[:, first]
    group = inverse[col]                                # entries sit in first columns only
    rep = (group_start[group + 1] - group_start[group]).astype(np.int64)
    idx = np.repeat(np.arange(len(col), dtype=np.int64), rep)
    within = np.arange(len(idx), dtype=np.int64) - np.repeat(np.cumsum(rep) - rep, rep)
    new_col = order[group_start[group[idx]] + within].astype(col.dtype)
    grown = np.zeros(len(rowptr), dtype=np.int64)
    np.cumsum(np.add.reduceat(np.concatenate([rep, [0]]), np.minimum(rowptr[:-1], len(rep)))
              * (np.diff(rowptr) > 0), out=grown[1:])
    return grown, new_col, val[idx], vecs


def vectorize_fasta(
    path: str,
    alphabet: Union[str, int, None],
    k: int,
    min_filter: int = 0,
    basis: Optional[Sequence[str]] = None,
    npz_out: Optional[str] = None,
    kmers_out: Optional[str] = None,
    sparse_npz_out: Optional[str] = None,
    timings: Optional[dict] = None,
    dense: Optional[bool] = None,
    compressed: bool = True,
    reference_pickle: bool = True,
) -> Dict[str, np.ndarray]:
    """FASTA -> the rule's outputs; optionally writes the ``.npz`` and the pickled KmerVec
    (``.kmers``) exactly as rules/kmerize.smk:132-142 does: the pickle names the reference's class path
    (``snekmer.vectorize.KmerVec``; `reference_pickle=False` names this package's), so scripts/cluster_cluster.py:53-63 reads it.  `sparse_npz_out` writes the sparse
    variant (io.save_npz_sparse: CSR counts, no dense matrix); without `npz_out` the dense
    N x |basis| float64 matrix is then never built.  `dense=False` skips it when nothing is written either; `compressed=False`
    writes the ``.npz`` members uncompressed (np.load reads both; the reference compresses, rules/kmerize.smk:132).  `timings` (optional
    dict) receives parse_s / gpu_s / decode_s / write_s."""
    t0 = time.perf_counter()
    ids, res, off, text_records = read_fasta_packed(path, with_records=True)
    if timings is not None:
        timings["parse_s"] = timings.get("parse_s", 0.0) + time.perf_counter() - t0
    want_dense = (bool(npz_out) or not sparse_npz_out) if dense is None else (dense or bool(npz_out))
    if text_records is not None and any(not s.isascii() and any(ord(c) > 255 for c in s) for _, s in text_records):
        # characters above latin-1 do not fit the packed bytes: vectorize_records carries them through by position
        out = vectorize_records(text_records, alphabet, k, min_filter=min_filter, basis=basis, dense=want_dense, timings=timings)
    else:
        out = vectorize_packed(ids, res, off, alphabet, k, min_filter=min_filter, basis=basis, dense=want_dense, timings=timings)
    t0 = time.perf_counter()
    if sparse_npz_out:
        save_npz_sparse(sparse_npz_out, out, compressed=compressed)
    if npz_out:
        save_npz(npz_out, dict(kmerlist=out["kmerlist"], ids=out["ids"], seqs=out["seqs"], vecs=out["vecs"], lengths=out["lengths"]),
                 compressed=compressed)
    if kmers_out:
        kmer = KmerVec(alphabet=alphabet, k=k)
        kmer.set_kmer_set(out["kmerlist"])
        with open(kmers_out, "wb") as f:
            dump_kmers(kmer, f, reference_pickle=reference_pickle)
    if timings is not None:
        timings["write_s"] = timings.get("write_s", 0.0) + time.perf_counter() - t0
    return out
