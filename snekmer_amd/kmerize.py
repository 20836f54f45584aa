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
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import engine
from .alphabet import FULL_ALPHABETS, build_lut
from .io import read_fasta, save_npz_sparse
from .vectorize import KmerVec, _restore_wide_chars


def vectorize_records(
    records: Sequence[Tuple[str, str]],
    alphabet: Union[str, int, None],
    k: int,
    min_filter: int = 0,
    basis: Optional[Sequence[str]] = None,
    dense: bool = True,
    ctx=None,
) -> Dict[str, np.ndarray]:
    from . import _hip

    ctx = ctx or _hip.default_context()
    lut = build_lut(alphabet)
    ids = [r[0] for r in records]
    raw = [str(r[1]) for r in records]
    n = len(raw)
    batch = engine.SeqBatch.from_strings(ctx, raw)

    # reduced strings (kmerize.smk:121-127)
    red_bytes, red_len = engine.recode(ctx, batch, lut)
    blob = red_bytes.tobytes()
    off = batch.h_offsets
    seqs = [
        _restore_wide_chars(s, blob[int(off[i]) : int(off[i]) + int(red_len[i])].decode("latin-1"))
        for i, s in enumerate(raw)
    ]

    # counts + observed basis
    csr = engine.count_csr(ctx, batch, lut, k, with_firstpos=True)
    b = engine.build_basis(ctx, csr, lut.nsym, k, stats=True, first_seen=True, postings=False)
    B = b.ncols
    bcodes = b.codes.download(B).astype(np.uint64)
    if basis is None:
        order = b.fs_order.download(B).astype(np.int64)          # columns in first-seen order
        total = b.total.download(B)
        keep = order[total[order] > min_filter]                 # kmerize.smk:102-104
        kmerlist = lut.decode(bcodes[keep], k)
        colmap = np.full(max(B, 1), 0xFFFFFFFF, dtype=np.uint32)
        colmap[keep] = np.arange(keep.size, dtype=np.uint32)
        ncols_out = int(keep.size)
    else:
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
    out = {
        "kmerlist": kmerlist if len(kmerlist) else np.array([], dtype=str),
        "ids": np.asarray(ids, dtype=str) if n else np.array([], dtype=str),
        "seqs": np.asarray(seqs, dtype=str) if n else np.array([], dtype=str),
        "lengths": np.asarray([len(s) for s in raw], dtype=np.int64),
    }
    if dense:
        vecs = engine.csr_to_dense(ctx, n, csr.rowptr, csr.colidx, csr.counts, ncols_out, colmap=d_colmap,
                                   presence=True, dtype=np.float64).download()
        out["vecs"] = vecs.reshape(max(n, 1), max(ncols_out, 1))[:n, :ncols_out]
    # additive: integer counts in the kmerlist column order, CSR
    rowptr, _, counts, _ = csr.host()
    colidx = csr.colidx.download(csr.nnz)
    newcol = colmap[colidx] if csr.nnz else np.zeros(0, dtype=np.uint32)
    keep_e = newcol != 0xFFFFFFFF
    row_of = np.repeat(np.arange(n), np.diff(rowptr))
    out["counts_rowptr"] = np.concatenate([[0], np.cumsum(np.bincount(row_of[keep_e], minlength=n))]).astype(np.int64)
    out["counts_col"] = newcol[keep_e].astype(np.uint32)
    out["counts_val"] = counts[keep_e].astype(np.uint32)
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
) -> Dict[str, np.ndarray]:
    """FASTA -> the rule's outputs; optionally writes the ``.npz`` and the pickled KmerVec
    (``.kmers``) exactly as rules/kmerize.smk:132-142 does.  `sparse_npz_out` writes the sparse
    variant (io.save_npz_sparse: CSR counts, no dense matrix); without `npz_out` the dense
    N x |basis| float64 matrix is then never built."""
    out = vectorize_records(read_fasta(path), alphabet, k, min_filter=min_filter, basis=basis,
                            dense=bool(npz_out) or not sparse_npz_out)
    if sparse_npz_out:
        save_npz_sparse(sparse_npz_out, out)
    if npz_out:
        np.savez_compressed(
            npz_out, kmerlist=out["kmerlist"], ids=out["ids"], seqs=out["seqs"], vecs=out["vecs"], lengths=out["lengths"]
        )
    if kmers_out:
        kmer = KmerVec(alphabet=alphabet, k=k)
        kmer.set_kmer_set(out["kmerlist"])
        with open(kmers_out, "wb") as f:
            pickle.dump(kmer, f)
    return out
