"""vectorize: k-mer vectors over a recoded alphabet, computed on the MI355X.

Public surface follows ``snekmer/vectorize.py`` (KmerBasis :18-125, KmerSet :135-169,
reduce :173-195, make_feature_matrix :201-221, KmerVec :224-345) so the Snakemake rule
bodies (``rules/kmerize.smk:67-142``, ``scripts/cluster_cluster.py:62-76``) can use these
classes unchanged; objects pickle with the same attribute names.  The per-record methods
are thin batch-of-one wrappers over the device path; the ``*_batch`` methods are the ones
meant for real workloads.  There is no CPU fallback: without libsnekmer_hip.so and a GPU
every numeric method raises ``snekmer_amd._hip.HipUnavailable``.
"""
import itertools
from typing import Dict, Generator, List, Optional, Sequence, Set, Union

import numpy as np

from ._version import __version__
from .alphabet import FULL_ALPHABETS, AlphabetLUT, build_lut, get_alphabet, get_alphabet_keys
from .utils import check_list, pack_sequences


def _ctx():
    from . import _hip

    return _hip.default_context()


# --------------------------------------------------------------------------------------
class KmerBasis:
    """Ordered k-mer basis and re-indexing of vectors into it (snekmer/vectorize.py:18-125)."""

    def __init__(self):
        self.basis = []
        self.basis_order = {}

    def set_basis(self, basis):
        if not check_list(basis):
            raise TypeError("`basis` input must be list or array-like.")
        self.basis = basis
        self.basis_order = {i: k for i, k in enumerate(basis)}

    def transform(self, vector, vector_basis):
        """(m, n) array in `vector_basis` column order -> (m, p) array in this basis' order;
        k-mers absent from `vector_basis` become zero columns.  Same errors as the reference:
        TypeError for a non-list basis, ValueError on a width mismatch, IndexError for 1-D input.
        The column lookup is a host-side string join (dict), the move itself a gather."""
        if not check_list(vector_basis):
            raise TypeError("`vector_basis` input must be list or array-like.")
        if not isinstance(vector, np.ndarray):
            vector = np.asarray(vector)
        try:
            vector_size = vector.shape[1]
        except IndexError:
            vector_size = len(vector)
        if vector_size != len(vector_basis):
            raise ValueError(
                "Vector and supplied basis shapes"
                " must match (vector shape ="
                f" {vector.shape}"
                " and len(vector_basis) ="
                f" {len(vector_basis)})."
            )
        where = {k: i for i, k in enumerate(vector_basis)}
        n_in = vector.shape[1]  # IndexError for 1-D input, as upstream
        none = 0xFFFFFFFF
        source = np.fromiter((where.get(self.basis_order[i], none) for i in range(len(self.basis))),
                             dtype=np.uint32, count=len(self.basis))
        return _gather_columns(vector, source)


def _gather_columns(matrix: np.ndarray, source: np.ndarray) -> np.ndarray:
    """out[:, p] = matrix[:, source[p]] (zero column where source[p] == 0xFFFFFFFF), moved on the
    device by skm_gather_columns.  Element types of 1/2/4/8 bytes are moved as raw words."""
    import ctypes as C

    rows, n_in = matrix.shape
    p_out = int(source.size)
    if matrix.dtype.itemsize not in (1, 2, 4, 8) or matrix.dtype.hasobject:
        raise NotImplementedError(f"KmerBasis.transform: unsupported element type {matrix.dtype}")
    out = np.zeros((rows, p_out), dtype=matrix.dtype)
    if rows == 0 or p_out == 0:
        return out
    ctx = _ctx()
    src = np.ascontiguousarray(matrix)
    d_in = ctx.to_device(src.view(np.uint8).reshape(-1)) if src.size else ctx.zeros(1, np.uint8)
    d_src = ctx.to_device(source)
    d_out = ctx.empty(out.nbytes, np.uint8)
    ctx.call("skm_gather_columns", C.c_int64(rows), C.c_int64(p_out), matrix.dtype.itemsize, C.c_void_p(d_in.ptr),
             C.c_int64(n_in), C.c_void_p(d_src.ptr), C.c_void_p(d_out.ptr))
    return d_out.download().view(matrix.dtype).reshape(rows, p_out)


def _generate(alphabet: Set[str], k: int):
    for c in itertools.product(alphabet, repeat=k):
        yield "".join(c)


class KmerSet:
    """Explicit k-mer list, or the full |alphabet|^k enumeration when `kmers` is None
    (snekmer/vectorize.py:135-169; the reference itself warns about the latter's size)."""

    def __init__(self, alphabet: Union[str, int], k: int, kmers: list = None):
        self.alphabet = alphabet
        self.k = k
        if kmers is None:
            self._kmerlist = list(_generate(get_alphabet_keys(alphabet), k))
        else:
            self._kmerlist = kmers

    @property
    def kmers(self):
        return iter(self._kmerlist)


# --------------------------------------------------------------------------------------
def _restore_wide_chars(original: str, reduced: str) -> str:
    if all(ord(c) < 256 for c in original):
        return reduced
    return "".join(o if ord(o) > 255 else r for o, r in zip(original, reduced))


_LUTS: Dict[tuple, AlphabetLUT] = {}


def _cached_lut(alphabet, mapping) -> AlphabetLUT:
    """One LUT per (alphabet, mapping object); a mapping that was edited in place (alphabet.register_alphabet)
    is recognised by its current table."""
    table = get_alphabet(alphabet, mapping)
    key = (str(alphabet), id(mapping), tuple(sorted((k, v) for k, v in dict(table).items() if k != "_keys")))
    lut = _LUTS.get(key)
    if lut is None:
        if len(_LUTS) > 64:
            _LUTS.clear()
        lut = _LUTS[key] = build_lut(alphabet, mapping)
    return lut


def reduce_batch(sequences: Sequence[str], alphabet: Union[str, int], mapping: dict = FULL_ALPHABETS) -> List[str]:
    """`reduce` for many sequences in one device call."""
    from . import engine

    seqs = [str(s) for s in sequences]
    lut = _cached_lut(alphabet, mapping)
    ctx = _ctx()
    data, off = pack_sequences(seqs)
    out, lens = engine.recode_host(ctx, data, off, lut)
    raw = out.tobytes()
    return [
        _restore_wide_chars(s, raw[int(off[i]) : int(off[i]) + int(lens[i])].decode("latin-1"))
        for i, s in enumerate(seqs)
    ]


def reduce(sequence: str, alphabet: Union[str, int], mapping: dict = FULL_ALPHABETS) -> str:
    """Recode one sequence: strip trailing '*', translate through the alphabet map; unmapped
    characters pass through (snekmer/vectorize.py:173-195)."""
    return reduce_batch([sequence], alphabet, mapping)[0]


def make_feature_matrix(vecs, min_filter=1, max_filter=1):
    """Ragged k-mer string lists -> (list of 0/1 float64 rows, sorted kmerlist kept iff total
    occurrences > min_filter) (snekmer/vectorize.py:201-221).  `max_filter` is unused upstream.

    Strings carry no alphabet, so a local one is derived from the characters present; the
    unique/filter/membership work then runs on the device."""
    import ctypes as C

    from . import engine

    rows = [np.asarray(v, dtype=str).ravel() for v in vecs]
    flat = np.concatenate(rows) if rows else np.array([], dtype=str)
    n = len(rows)
    if flat.size == 0:
        return [np.zeros(0) for _ in rows], np.array([], dtype=str)
    k = flat.dtype.itemsize // 4
    lengths = np.char.str_len(flat)
    if not np.all(lengths == k):
        raise NotImplementedError("make_feature_matrix: k-mers of differing lengths are unsupported")
    chars = flat.view(np.uint32).reshape(flat.size, k)
    letters = np.unique(chars)
    nsym = int(letters.size)
    if nsym**k >= 2**64:
        raise NotImplementedError("make_feature_matrix: k-mer space exceeds 64-bit codes")
    ranks = np.searchsorted(letters, chars).astype(np.uint64)
    codes = np.zeros(flat.size, dtype=np.uint64)
    for j in range(k):
        codes = codes * np.uint64(nsym) + ranks[:, j]
    bits = 32 if nsym**k < 2**32 else 64
    ctx = _ctx()
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=rowptr[1:])
    csr = engine.CountsCSR(
        ctx, n, int(flat.size), bits, ctx.to_device(rowptr),
        ctx.to_device(codes.astype(np.uint32 if bits == 32 else np.uint64)),
        ctx.to_device(np.ones(flat.size, dtype=np.uint32)), None,
    )
    basis = engine.build_basis(ctx, csr, nsym, k, stats=True, postings=False)
    total = basis.total.download(basis.ncols)
    bcodes = basis.codes.download(basis.ncols).astype(np.uint64)
    keep = total > min_filter
    colmap = np.full(basis.ncols, 0xFFFFFFFF, dtype=np.uint32)
    colmap[keep] = np.arange(int(keep.sum()), dtype=np.uint32)
    nk = int(keep.sum())
    dense = engine.csr_to_dense(ctx, n, csr.rowptr, csr.colidx, csr.counts, nk, colmap=ctx.to_device(colmap),
                                presence=True).download()
    dense = dense.reshape(max(n, 1), max(nk, 1))[:n, :nk]
    # decode kept codes with the local alphabet
    kept = bcodes[keep]
    out_chars = np.empty((kept.size, k), dtype=np.uint32)
    rem = kept.copy()
    for j in range(k - 1, -1, -1):
        out_chars[:, j] = letters[(rem % np.uint64(nsym)).astype(np.intp)]
        rem //= np.uint64(nsym)
    kmerlist = out_chars.view(f"<U{k}").ravel() if kept.size else np.array([], dtype=flat.dtype)
    return [dense[i].copy() for i in range(n)], kmerlist


# --------------------------------------------------------------------------------------
class KmerVec:
    """k-mer vectoriser for one (alphabet, k) (snekmer/vectorize.py:224-345)."""

    def __init__(self, alphabet: Union[str, int], k: int):
        self.alphabet = alphabet
        self.k = k
        self.char_set = get_alphabet_keys(alphabet)
        self.vector = None
        self.basis = KmerBasis()
        self.snekmer_version = __version__

    # The LUT is derived state: built once per object and kept OUT of the pickle, so that .kmers files keep
    # the reference's attribute set (snekmer/vectorize.py:225-231).
    def _lut(self) -> AlphabetLUT:
        lut = self.__dict__.get("_lut_cache")
        if lut is None or self.__dict__.get("_lut_for") != self.alphabet:
            lut = build_lut(self.alphabet)
            self.__dict__["_lut_cache"] = lut
            self.__dict__["_lut_for"] = self.alphabet
        return lut

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_lut_cache", None)
        state.pop("_lut_for", None)
        return state

    def set_kmer_set(self, kmer_set=list()):
        self.kmer_set = KmerSet(self.alphabet, self.k, kmer_set)
        self.basis.set_basis(kmer_set)

    def _kmer_gen(self, sequence: str) -> Generator[str, None, None]:
        """Valid k-mers of an already-reduced string, window order (snekmer/vectorize.py:239-249):
        a window counts iff every character is a class letter."""
        lut = self._lut()
        # identity translate: `sequence` is already in class-letter space
        ident = AlphabetLUT(np.arange(256, dtype=np.uint8), _rank_of_letters(lut), lut.letters)
        for kmer in self._windows_batch([str(sequence)], ident, strip=False)[0]:
            yield str(kmer)

    @staticmethod
    def _kmer_gen_str(sequence: str, k: int) -> Generator[str, None, None]:
        for n in range(0, len(sequence) - k + 1):
            yield sequence[n : n + k]

    def _windows_batch(self, seqs: Sequence[str], lut: AlphabetLUT, strip: bool = True) -> List[np.ndarray]:
        from . import engine

        ctx = _ctx()
        if not strip:
            # trailing '*' must survive as an (invalid) character: shield it from the strip
            seqs = [s + "\x00" if s.endswith("*") else s for s in seqs]
        data, off = pack_sequences(seqs)
        # small batches (the per-record calls of an unchanged rule body) go through a per-context arena:
        # one upload, two launches, one download, no allocation
        codes, nwin, bits = engine.kmer_codes_host(ctx, data, off, lut, self.k)
        sentinel = np.iinfo(codes.dtype).max
        n = len(off) - 1
        if n == 1:
            w = codes[: int(nwin[0])]
            return [lut.decode(w[w != sentinel], self.k)]
        # every record's valid windows decoded in ONE pass, then cut per record
        slot = np.arange(codes.size, dtype=np.int64)
        rec = np.searchsorted(off, slot, side="right") - 1
        live = (slot - off[rec] < nwin[rec]) & (codes != sentinel)
        words = lut.decode(codes[live], self.k)
        per = np.bincount(rec[live], minlength=n)
        cuts = np.cumsum(per)[:-1]
        parts = np.split(words, cuts) if words.size else [words[:0] for _ in range(n)]
        empty = np.array([], dtype=str)  # '<U1', shape (0,): what np.array([], dtype=str) gives upstream
        return [p if p.size else empty for p in parts]

    def reduce_vectorize_batch(self, sequences: Sequence[str]) -> List[np.ndarray]:
        """`reduce_vectorize` for many sequences in one device call."""
        return self._windows_batch([str(s) for s in sequences], self._lut())

    def reduce_vectorize(self, sequence: str) -> np.ndarray:
        """Recode + list the valid k-mers of one sequence as a numpy str array, window order,
        duplicates kept (snekmer/vectorize.py:292-328)."""
        return self.reduce_vectorize_batch([sequence])[0]

    def vectorize(self, sequence: str) -> np.ndarray:
        """k-mer count vector of an already-reduced sequence over ``self.kmer_set`` order.

        Upstream's method of this name raises KeyError on its first iteration
        (snekmer/vectorize.py:282-285), so nothing can pin this; it implements the documented
        intent (:259-271).  Parity: unpinned."""
        from . import engine

        lut = self._lut()
        ident = AlphabetLUT(np.arange(256, dtype=np.uint8), _rank_of_letters(lut), lut.letters)
        seq = str(sequence)
        if seq.endswith("*"):  # already-reduced input: a trailing '*' is an ordinary invalid character
            seq += "\x00"
        ctx = _ctx()
        csr = engine.count_csr(ctx, engine.SeqBatch.from_strings(ctx, [seq]), ident, self.k)
        _, codes, counts, _ = csr.host()
        codes = codes.astype(np.uint64)
        kmers = [str(w) for w in self.kmer_set.kmers]
        want, ok = ident.encode(kmers, self.k)
        if codes.size == 0:
            return np.zeros(len(kmers), dtype=np.int64)
        pos = np.clip(np.searchsorted(codes, want), 0, codes.size - 1)
        hit = ok & (codes[pos] == want)
        return np.where(hit, counts[pos], 0).astype(np.int64)

    def count_batch(self, sequences: Sequence[str], with_firstpos: bool = False):
        """Device-resident per-sequence (code, count) lists: see engine.count_csr."""
        from . import engine

        ctx = _ctx()
        batch = engine.SeqBatch.from_strings(ctx, [str(s) for s in sequences])
        return engine.count_csr(ctx, batch, self._lut(), self.k, with_firstpos=with_firstpos)

    def harmonize(self, record, kmerlist):
        return self.basis.transform(record, kmerlist)


def _rank_of_letters(lut: AlphabetLUT) -> np.ndarray:
    rank = np.full(256, 0xFF, dtype=np.uint8)
    for i, ch in enumerate(lut.letters):
        rank[ord(ch)] = i
    return rank
