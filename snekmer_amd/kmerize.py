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
from .io import read_fasta, read_fasta_packed, save_npz_sparse
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
        # a k-mer listed twice in the basis keeps only its last column non-zero upstream as well
        colmap[pos_c[hit]] = np.nonzero(hit)[0].astype(np.uint32)
        ncols_out = int(len(kmerlist))
        if len(set(kmerlist.tolist())) != len(kmerlist):
            raise NotImplementedError("explicit basis with repeated k-mers is unsupported")
        d_colmap = ctx.to_device(colmap)
    # additive: integer counts in the kmerlist column order, CSR
    c_rowptr, c_col, c_val = engine.csr_remap_columns(ctx, csr, d_colmap, B)
    vecs = None
    if dense:
        vecs = engine.csr_to_dense(ctx, n, csr.rowptr, csr.colidx, csr.counts, ncols_out, colmap=d_colmap,
                                   presence=True, dtype=np.float64).download()
        vecs = vecs.reshape(max(n, 1), max(ncols_out, 1))[:n, :ncols_out]
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
) -> Dict[str, np.ndarray]:
    """FASTA -> the rule's outputs; optionally writes the ``.npz`` and the pickled KmerVec
    (``.kmers``) exactly as rules/kmerize.smk:132-142 does.  `sparse_npz_out` writes the sparse
    variant (io.save_npz_sparse: CSR counts, no dense matrix); without `npz_out` the dense
    N x |basis| float64 matrix is then never built.  `dense=False` skips it when nothing is written either; `compressed=False`
    writes the ``.npz`` members uncompressed (np.load reads both; the reference compresses, rules/kmerize.smk:132).  `timings` (optional
    dict) receives parse_s / gpu_s / decode_s / write_s."""
    t0 = time.perf_counter()
    ids, res, off = read_fasta_packed(path)
    if timings is not None:
        timings["parse_s"] = timings.get("parse_s", 0.0) + time.perf_counter() - t0
    out = vectorize_packed(ids, res, off, alphabet, k, min_filter=min_filter, basis=basis,
                           dense=(bool(npz_out) or not sparse_npz_out) if dense is None else (dense or bool(npz_out)),
                           timings=timings)
    t0 = time.perf_counter()
    if sparse_npz_out:
        save_npz_sparse(sparse_npz_out, out, compressed=compressed)
    if npz_out:
        (np.savez_compressed if compressed else np.savez)(
            npz_out, kmerlist=out["kmerlist"], ids=out["ids"], seqs=out["seqs"], vecs=out["vecs"], lengths=out["lengths"]
        )
    if kmers_out:
        kmer = KmerVec(alphabet=alphabet, k=k)
        kmer.set_kmer_set(out["kmerlist"])
        with open(kmers_out, "wb") as f:
            pickle.dump(kmer, f)
    if timings is not None:
        timings["write_s"] = timings.get("write_s", 0.0) + time.perf_counter() - t0
    return out
