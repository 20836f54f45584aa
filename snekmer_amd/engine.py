"""engine: batched device operations of the hot path, composed from the C-ABI entry points.

Data stays in HBM between stages:

    SeqBatch --count_csr--> CountsCSR --build_basis--> Basis (+ column ids, postings)
                                         |                         |
                                         +------ cosine_matrix ----+--> float32 [rows x M] in HBM

The reference does each of these per sequence in Python (snekmer/rules/kmerize.smk:89-129,
snekmer/rules/learn.smk:359-383, snekmer/rules/apply.smk:188-206,278-289); here each stage is
one call per batch.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _hip
from .alphabet import AlphabetLUT

_p = C.c_void_p
_i64 = C.c_int64


def _ptr(arr) -> _p:
    if arr is None:
        return _p(None)
    if isinstance(arr, _hip.DeviceArray):
        return _p(arr.ptr)
    if isinstance(arr, np.ndarray):
        return arr.ctypes.data_as(_p)
    return _p(int(arr))


def key_bits(nsym: int, k: int) -> int:
    return max(1, int(nsym**k - 1).bit_length())


class SeqBatch:
    """Packed sequences resident on the device (bytes + int64 offsets[n+1])."""

    def __init__(self, ctx: _hip.Context, residues: np.ndarray, offsets: np.ndarray):
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or offsets.size < 1 or offsets[0] != 0 or np.any(np.diff(offsets) < 0):
            raise ValueError("offsets must start at 0 and be non-decreasing")
        if int(offsets[-1]) != residues.size:
            raise ValueError("offsets[-1] must equal the number of residues")
        self.ctx = ctx
        self.n = int(offsets.size - 1)
        self.total = int(residues.size)
        self.h_offsets = offsets
        # bound on the longest sequence: lets the count stage size every launch without asking the device (skm_count_csr)
        self.max_len = int(np.diff(offsets).max()) if self.n else 0
        # 64 spare bytes keep 16-byte vector loads of the tail in bounds
        self.d_seq = ctx.zeros(self.total + 64, np.uint8)
        if self.total:
            ctx._h2d(self.d_seq.ptr, residues)
        self.d_off = ctx.to_device(offsets)
        self.ready = None        # (context, event slot) of an upload that a consumer on another context must wait for
        self.on_consumed = None  # called with (context, event slot) by a pipeline that has queued the batch's last reader

    @classmethod
    def from_strings(cls, ctx, seqs: Sequence[str]) -> "SeqBatch":
        from .utils import pack_sequences

        data, off = pack_sequences(seqs)
        return cls(ctx, data, off)

    @classmethod
    def _view(cls, ctx, n, total, h_offsets, max_len, d_seq, d_off) -> "SeqBatch":
        self = cls.__new__(cls)
        self.ctx, self.n, self.total, self.h_offsets, self.max_len = ctx, n, total, h_offsets, max_len
        self.d_seq, self.d_off = d_seq, d_off
        self.ready = self.on_consumed = None
        return self


class BatchUploader:
    """A stream of batches that arrives from the host (every job of the reference starts from a file,
    snekmer/rules/kmerize.smk:89-129): `slots` recycled pairs of device buffers and pinned staging buffers on a COPY
    context of its own; `upload(residues, offsets)` packs a batch into the next staging buffer and queues two
    asynchronous copies - no allocation, no fill, no host wait per batch - and returns a SeqBatch whose `ready` event a
    consumer on another context waits for ON THE DEVICE (engine.OverlappedPipeline.prefetch does).  A slot is refilled
    only behind the event the consumer recorded after queuing the slot's last reader (`on_consumed`), and its staging
    buffer only once the copy out of it has run.  The caller hands every batch to a pipeline before `slots` further
    uploads come round to its slot (bench.py keeps two uploads ahead of the vectorize with three slots): a batch that no
    pipeline has taken has no reader to wait for and would be overwritten.

        up = BatchUploader(ctx, max_residues, max_sequences, slots=3)
        pipe.prefetch(up.upload(res0, off0))
        for res, off in files:
            out = pipe.step(up.upload(res, off))     # the copy of batch i + 1 runs beside batch i's kernels
    """

    PAD = 64  # zero bytes behind the residues: 16-byte vector loads of the tail stay in bounds and read zeros

    def __init__(self, ctx: _hip.Context, max_residues: int, max_sequences: int, slots: int = 3):
        if not 2 <= slots <= _hip.EVENT_SLOTS:
            raise ValueError(f"slots: 2..{_hip.EVENT_SLOTS}")
        self.ctx = ctx  # the device (and the context the batches nominally live on)
        self.copy = _hip.Context(ctx.device)
        self.cap_res, self.cap_n = int(max_residues), int(max_sequences)
        self.slots = []
        for _ in range(slots):
            h_seq = self.copy.host_alloc(self.cap_res + self.PAD)
            h_off = self.copy.host_alloc(8 * (self.cap_n + 1)).view(np.int64)
            self.slots.append({"d_seq": self.copy.empty(self.cap_res + self.PAD, np.uint8), "d_off": self.copy.empty(self.cap_n + 1, np.int64),
                               "h_seq": h_seq, "h_off": h_off, "consumed": None, "used": False})
        self.nxt = 0
        self.host_waits = 0  # times upload() had to wait for a staging buffer's previous copy (0 in a well-fed stream)

    def upload(self, residues: np.ndarray, offsets: np.ndarray) -> SeqBatch:
        residues = np.asarray(residues, dtype=np.uint8)
        offsets = np.asarray(offsets, dtype=np.int64)
        n, total = int(offsets.size - 1), int(residues.size)
        if offsets.ndim != 1 or offsets.size < 1 or offsets[0] != 0 or int(offsets[-1]) != total or np.any(np.diff(offsets) < 0):
            raise ValueError("offsets must start at 0, be non-decreasing and end at the number of residues")
        if total > self.cap_res or n > self.cap_n:
            raise ValueError(f"batch of {n} sequences / {total} residues exceeds the uploader's capacity ({self.cap_n} / {self.cap_res})")
        j = self.nxt
        self.nxt = (j + 1) % len(self.slots)
        sl = self.slots[j]
        if sl["used"] and not self.copy.event_done(j):  # the staging buffer's previous copy has not run yet
            self.host_waits += 1
            self.copy.sync()
        if sl["consumed"] is not None:  # the device buffers' last reader (another context's kernels)
            self.copy.wait_event(*sl["consumed"])
            sl["consumed"] = None
        sl["h_seq"][:total] = residues
        sl["h_seq"][total:total + self.PAD] = 0
        sl["h_off"][: n + 1] = offsets
        self.copy.h2d_async(sl["d_seq"].ptr, sl["h_seq"], total + self.PAD)
        self.copy.h2d_async(sl["d_off"].ptr, sl["h_off"], 8 * (n + 1))
        self.copy.record_event(j)
        sl["used"] = True
        batch = SeqBatch._view(self.ctx, n, total, offsets, int(np.diff(offsets).max()) if n else 0, sl["d_seq"], sl["d_off"])
        batch.ready = (self.copy, j)

        def consumed(ctx, slot, sl=sl):
            sl["consumed"] = (ctx, slot)

        batch.on_consumed = consumed
        return batch

    def close(self):
        self.copy.sync()
        self.slots = []
        self.copy.close()


class CountsCSR:
    """Per-sequence (code, count) lists on the device; codes ascending within a row."""

    def __init__(self, ctx, n, nnz, code_bits, rowptr, codes, counts, firstpos):
        self.ctx, self.n, self.code_bits = ctx, n, code_bits
        self.rowptr, self.codes, self.counts, self.firstpos = rowptr, codes, counts, firstpos
        self.nnz = nnz
        self.colidx: Optional[_hip.DeviceArray] = None
        # True when colidx carries 0xFFFFFFFF for k-mers of one row only (build_basis(elide_singletons=True)):
        # such a CSR is only meaningful as the X side of the SQUARE cosine against its own postings
        self.elided = False

    @property
    def code_dtype(self):
        return np.uint32 if self.code_bits == 32 else np.uint64

    # The entry count may live on the device only (vectorize_fused leaves it in rowptr[n] and reads no result
    # back): it is fetched, once, when the host first asks for it.
    @property
    def nnz(self) -> int:
        if self._nnz is None:
            self._nnz = int(self.rowptr.download(1, offset=self.n)[0])
        return self._nnz

    @nnz.setter
    def nnz(self, value):
        self._nnz = None if value is None else int(value)

    def host(self):
        """(rowptr, codes, counts, firstpos|None) as numpy arrays."""
        fp = self.firstpos.download(self.nnz) if self.firstpos is not None else None
        return (
            self.rowptr.download(self.n + 1),
            self.codes.download(self.nnz),
            self.counts.download(self.nnz),
            fp,
        )


class Basis:
    """Observed k-mer basis of one CountsCSR plus its column-major copy (postings)."""

    def __init__(self):
        self._ncols = 0
        self.d_ncols = None  # device int64 written by skm_vectorize_csr
        self.codes = self.df = self.total = self.firstkey = self.fs_order = None
        self.colptr = self.post = None
        # posting words: 64 = row | count << 32; 32 = row | min(count, 255) << 24 with the real counts of
        # saturated postings in `postcnt` (include/snekmer_hip.h, SKM_BASIS_POST32)
        self.post_bits = 64
        self.postcnt = None

    @property
    def ncols(self) -> int:
        """Number of basis columns; fetched from the device on first use after a fused vectorize."""
        if self._ncols is None:
            self._ncols = int(self.d_ncols.download(1)[0])
        return self._ncols

    @ncols.setter
    def ncols(self, value):
        self._ncols = None if value is None else int(value)

    def ncols_hint(self) -> int:
        """The column count if the host already knows it, else an upper bound (the kernels only validate it)."""
        return self._ncols if self._ncols is not None else int(self.colptr.size - 1)


def recode(ctx: _hip.Context, batch: SeqBatch, lut: AlphabetLUT) -> Tuple[np.ndarray, np.ndarray]:
    """a3: translated bytes (same layout as the input) and stripped lengths, on the host."""
    d_out = ctx.empty(batch.total + 64, np.uint8)
    d_len = ctx.empty(max(batch.n, 1), np.int32)
    ctx.call("skm_recode", _ptr(lut.translate), _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n), _ptr(d_out), _ptr(d_len))
    return d_out.download(batch.total), d_len.download(batch.n)


def kmer_codes(ctx, batch: SeqBatch, lut: AlphabetLUT, k: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """a5/a6: (codes per window slot, windows per sequence, code_bits) on the host."""
    bits = lut.code_bits(k)
    dt = np.uint32 if bits == 32 else np.uint64
    d_codes = ctx.empty(batch.total + 1, dt)
    d_nwin = ctx.zeros(max(batch.n, 1), np.int32)
    ctx.call(
        "skm_kmer_codes", _ptr(lut.rank), lut.nsym, k, bits, _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n),
        _ptr(d_codes), _ptr(d_nwin),
    )
    return d_codes.download(batch.total), d_nwin.download(batch.n), bits


ARENA_RESIDUES = 1 << 16  # batches up to this many residues / 4096 records use the per-context arena
ARENA_RECORDS = 4096


class _Arena:
    """Persistent staging of the small-batch path in pinned host memory that the device addresses directly
    (skm_host_alloc): [offsets | residues] are written by the host and read by the kernels over PCIe, [window counts |
    codes] are written by the kernels and read by the host.  A call is the launches plus one stream synchronise: no
    allocation and no explicit copy."""

    def __init__(self, ctx):
        self.up_bytes = 8 * (ARENA_RECORDS + 1) + ARENA_RESIDUES + 64
        self.h_up = ctx.host_alloc(self.up_bytes)
        self.down_bytes = 4 * ARENA_RECORDS + 8 * (ARENA_RESIDUES + 1) + 64
        self.h_down = ctx.host_alloc(self.down_bytes)
        self.up_ptr, self.down_ptr = self.h_up.ctypes.data, self.h_down.ctypes.data


def recode_host(ctx, residues: np.ndarray, offsets: np.ndarray, lut: AlphabetLUT):
    """a3 from host arrays to host arrays: (translated bytes in the input layout, stripped lengths); small inputs
    through the context's arena."""
    n, total = int(offsets.size - 1), int(residues.size)
    if n > ARENA_RECORDS or total > ARENA_RESIDUES or n == 0:
        return recode(ctx, SeqBatch(ctx, residues, offsets), lut)
    arena = ctx.__dict__.get("_arena")
    if arena is None:
        arena = ctx.__dict__["_arena"] = _Arena(ctx)
    off_bytes = 8 * (n + 1)
    seq_at = (off_bytes + 15) // 16 * 16
    arena.h_up[:off_bytes] = np.ascontiguousarray(offsets, dtype=np.int64).view(np.uint8)
    arena.h_up[seq_at : seq_at + total] = residues
    arena.h_up[seq_at + total : seq_at + total + 16] = 0
    len_bytes = (4 * n + 15) // 16 * 16
    ctx.call("skm_recode", _ptr(lut.translate), _p(arena.up_ptr + seq_at), _p(arena.up_ptr), _i64(n),
             _p(arena.down_ptr + len_bytes), _p(arena.down_ptr))
    ctx.sync()
    got = arena.h_down[: len_bytes + total]
    return got[len_bytes : len_bytes + total].copy(), got[: 4 * n].view(np.int32).copy()


def kmer_codes_host(ctx, residues: np.ndarray, offsets: np.ndarray, lut: AlphabetLUT, k: int):
    """a5/a6 from host arrays to host arrays: (codes per window slot, windows per sequence, code_bits).
    Small inputs use the context's arena (pinned zero-copy staging: no allocation, no copy call); larger ones the general path."""
    n, total = int(offsets.size - 1), int(residues.size)
    if n > ARENA_RECORDS or total > ARENA_RESIDUES or n == 0:
        batch = SeqBatch(ctx, residues, offsets)
        return kmer_codes(ctx, batch, lut, k)
    arena = ctx.__dict__.get("_arena")
    if arena is None:
        arena = ctx.__dict__["_arena"] = _Arena(ctx)
    bits = lut.code_bits(k)
    cb = bits // 8
    off_bytes = 8 * (n + 1)
    seq_at = (off_bytes + 15) // 16 * 16
    arena.h_up[:off_bytes] = np.ascontiguousarray(offsets, dtype=np.int64).view(np.uint8)
    arena.h_up[seq_at : seq_at + total] = residues
    arena.h_up[seq_at + total : seq_at + total + 16] = 0
    nwin_bytes = (4 * n + 15) // 16 * 16
    d_nwin, d_codes = arena.down_ptr, arena.down_ptr + nwin_bytes
    ctx.call("skm_kmer_codes", _ptr(lut.rank), lut.nsym, k, bits, _p(arena.up_ptr + seq_at), _p(arena.up_ptr), _i64(n),
             _p(d_codes), _p(d_nwin))
    ctx.sync()
    got = arena.h_down[: nwin_bytes + cb * total]
    nwin = got[: 4 * n].view(np.int32).copy()
    codes = got[nwin_bytes : nwin_bytes + cb * total].view(np.uint32 if bits == 32 else np.uint64).copy()
    return codes, nwin, bits


def count_csr(ctx, batch: SeqBatch, lut: AlphabetLUT, k: int, with_firstpos: bool = False,
              out: Optional[CountsCSR] = None) -> CountsCSR:
    """a12: per-sequence k-mer counts.  Pass `out` (from a previous call on an equally sized
    batch) to reuse its device buffers."""
    bits = lut.code_bits(k)
    dt = np.uint32 if bits == 32 else np.uint64
    cap = batch.total + 1
    if out is None or out.codes.size < cap or out.code_bits != bits or out.rowptr.size < batch.n + 1:
        rowptr = ctx.empty(batch.n + 1, np.int64)
        codes = ctx.empty(cap, dt)
        counts = ctx.empty(cap, np.uint32)
        firstpos = ctx.empty(cap, np.uint32) if with_firstpos else None
        out = CountsCSR(ctx, batch.n, 0, bits, rowptr, codes, counts, firstpos)
    elif with_firstpos and out.firstpos is None:
        out.firstpos = ctx.empty(cap, np.uint32)
    nnz = _i64(0)
    ctx.call(
        "skm_count_csr", _ptr(lut.rank), lut.nsym, k, bits, _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n),
        _i64(batch.total), _i64(batch.max_len), _i64(cap), _ptr(out.rowptr), _ptr(out.codes), _ptr(out.counts),
        _ptr(out.firstpos if with_firstpos else None), C.byref(nnz),
    )
    out.n, out.nnz = batch.n, int(nnz.value)
    return out


def build_basis(ctx, csr: CountsCSR, nsym: int, k: int, stats: bool = False, first_seen: bool = False,
                postings: bool = True, out: Optional[Basis] = None, elide_singletons: bool = False,
                post32: bool = False) -> Basis:
    """a11: observed basis (ascending codes), column ids for `csr`, optional stats / first-seen
    order / postings.  With `elide_singletons` (cosine-only use) entries of k-mers that occur in a
    single sequence get colidx 0xFFFFFFFF and no posting (see include/snekmer_hip.h).  `post32` asks
    for 4-byte posting words (fewer than 2^24 rows, no stats): half the bytes the cosine kernels gather."""
    nnz = csr.nnz
    b = out or Basis()
    cap = max(nnz, 1)
    dt = csr.code_dtype

    def need(cur, size, dtype):
        if cur is None or cur.size < size or cur.dtype != np.dtype(dtype):
            return ctx.empty(size, dtype)
        return cur

    b.codes = need(b.codes, cap, dt)
    csr.colidx = need(csr.colidx, cap, np.uint32)
    if stats:
        b.df = need(b.df, cap, np.uint32)
        b.total = need(b.total, cap, np.uint64)
    if first_seen:
        if csr.firstpos is None:
            raise ValueError("first_seen=True needs count_csr(..., with_firstpos=True)")
        b.firstkey = need(b.firstkey, cap, np.uint64)
        b.fs_order = need(b.fs_order, cap, np.uint32)
    post32 = bool(post32 and postings and not stats and csr.n < (1 << 24))
    if postings:
        b.colptr = need(b.colptr, cap + 1, np.uint32)
        b.post = need(b.post, cap, np.uint32 if post32 else np.uint64)
        b.postcnt = need(b.postcnt, cap, np.uint32) if post32 else None
    b.post_bits = 32 if post32 else 64
    ncols = _i64(0)
    flags = (1 if elide_singletons else 0) | (2 if post32 else 0)
    ctx.call(
        "skm_basis_build", csr.code_bits, key_bits(nsym, k), flags, _i64(csr.n), _i64(nnz), _ptr(csr.rowptr), _ptr(csr.codes),
        _ptr(csr.counts), _ptr(csr.firstpos if first_seen else None), C.byref(ncols), _ptr(b.codes), _ptr(csr.colidx),
        _ptr(b.df if stats else None), _ptr(b.total if stats else None), _ptr(b.firstkey if first_seen else None),
        _ptr(b.fs_order if first_seen else None), _ptr(b.colptr if postings else None),
        _ptr(b.post if postings else None), _ptr(b.postcnt if post32 else None),
    )
    b.ncols = int(ncols.value)
    csr.elided = bool(elide_singletons)
    return b


def vectorize_fused(ctx, batch: SeqBatch, lut: AlphabetLUT, k: int, csr: Optional[CountsCSR] = None,
                    basis: Optional[Basis] = None, rnorm=None):
    """count_csr + build_basis(elide_singletons, postings) + row_norms in one call that reads no result back
    (skm_vectorize_csr): sizes that depend on the data (entry count, number of columns) stay on the device and are
    fetched lazily by `CountsCSR.nnz` / `Basis.ncols`.  One host wait per call remains inside the library (the
    sequences' size-class histogram, awaited behind the classification kernel: include/snekmer_hip.h), so the host
    runs at most one step ahead of the device.  Returns (csr, basis, rnorm); pass the previous ones to reuse
    their buffers."""
    if batch.n < 1 or batch.total < 1:
        raise ValueError("vectorize_fused needs a non-empty batch")
    bits = lut.code_bits(k)
    dt = np.uint32 if bits == 32 else np.uint64
    cap = batch.total + 1
    if csr is None or csr.codes.size < cap or csr.code_bits != bits or csr.rowptr.size < batch.n + 1:
        csr = CountsCSR(ctx, batch.n, None, bits, ctx.empty(batch.n + 1, np.int64), ctx.empty(cap, dt),
                        ctx.empty(cap, np.uint32), None)
    if csr.colidx is None or csr.colidx.size < cap:
        csr.colidx = ctx.empty(cap, np.uint32)
    b = basis or Basis()
    if b.codes is None or b.codes.size < cap or b.codes.dtype != np.dtype(dt):
        b.codes = ctx.empty(cap, dt)
    if b.colptr is None or b.colptr.size < cap + 1:
        b.colptr = ctx.empty(cap + 1, np.uint32)
    if b.post is None or b.post.size < cap or b.post.dtype != np.dtype(np.uint64):
        b.post = ctx.empty(cap, np.uint64)
    if b.d_ncols is None:
        b.d_ncols = ctx.zeros(1, np.int64)
    b.post_bits, b.postcnt = 64, None
    rn = rnorm if rnorm is not None and rnorm.size >= batch.n else ctx.empty(batch.n + 4, np.float32)
    ctx.call(
        "skm_vectorize_csr", _ptr(lut.rank), lut.nsym, k, bits, _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n),
        _i64(batch.total), _i64(batch.max_len), _i64(cap), _ptr(csr.rowptr), _ptr(csr.codes), _ptr(csr.counts), _ptr(b.codes), _ptr(csr.colidx),
        _ptr(b.colptr), _ptr(b.post), _ptr(rn), _ptr(None), _ptr(b.d_ncols),
    )
    csr.n, csr.nnz, csr.elided = batch.n, None, True
    b.ncols = None
    return csr, b, rn


def vectorize_counts(ctx, batch: SeqBatch, lut: AlphabetLUT, k: int, csr: Optional[CountsCSR] = None, rnorm=None):
    """The count stage of vectorize_fused alone (skm_vectorize_csr without basis outputs): CSR counts + row norms, no
    result read back.  Returns (csr, rnorm); `csr.colidx` is left as it was (no basis has been built)."""
    if batch.n < 1 or batch.total < 1:
        raise ValueError("vectorize_counts needs a non-empty batch")
    bits = lut.code_bits(k)
    dt = np.uint32 if bits == 32 else np.uint64
    cap = batch.total + 1
    if csr is None or csr.codes.size < cap or csr.code_bits != bits or csr.rowptr.size < batch.n + 1:
        csr = CountsCSR(ctx, batch.n, None, bits, ctx.empty(batch.n + 1, np.int64), ctx.empty(cap, dt),
                        ctx.empty(cap, np.uint32), None)
    rn = rnorm if rnorm is not None and rnorm.size >= batch.n else ctx.empty(batch.n + 4, np.float32)
    ctx.call(
        "skm_vectorize_csr", _ptr(lut.rank), lut.nsym, k, bits, _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n),
        _i64(batch.total), _i64(batch.max_len), _i64(cap), _ptr(csr.rowptr), _ptr(csr.codes), _ptr(csr.counts), _ptr(None), _ptr(None),
        _ptr(None), _ptr(None), _ptr(rn), _ptr(None), _ptr(None),
    )
    csr.n, csr.nnz, csr.elided = batch.n, None, False
    return csr, rn


def row_norms(ctx, n: int, rowptr, counts, out=None) -> _hip.DeviceArray:
    rn = out if out is not None and out.size >= max(n, 1) else ctx.empty(max(n, 1) + 4, np.float32)
    ctx.call("skm_row_norms_csr", _i64(n), _ptr(rowptr), _ptr(counts), _ptr(rn), _ptr(None))
    return rn


def row_normsq(ctx, n: int, rowptr, counts) -> _hip.DeviceArray:
    """Exact squared row norms (uint64) of a CSR count matrix."""
    nsq = ctx.empty(max(n, 1), np.uint64)
    ctx.call("skm_row_norms_csr", _i64(n), _ptr(rowptr), _ptr(counts), _ptr(None), _ptr(nsq))
    return nsq


def row_norms_i8(ctx, n: int, kdim: int, mat, out=None) -> _hip.DeviceArray:
    """1/||row|| of an int8 matrix [n x kdim] (the operand layout of cosine_dense_i8)."""
    rn = out if out is not None and out.size >= max(n, 1) else ctx.empty(max(n, 1) + 4, np.float32)
    ctx.call("skm_row_norms_i8", _i64(n), _i64(kdim), _ptr(mat), _ptr(rn), _ptr(None))
    return rn


def dense_to_csr(ctx, dense, n: int, ncols: int, ld: int, cap_entries: int) -> CountsCSR:
    """Sparse view of a dense count matrix (uint16 / uint32 / int8 cells): CountsCSR whose `codes` are
    the non-zero column ids (ascending per row) and `counts` their values (skm_dense_to_csr)."""
    code = {np.dtype(np.uint16): 0, np.dtype(np.uint32): 1, np.dtype(np.int8): 2}[np.dtype(dense.dtype)]
    rowptr = ctx.empty(n + 1, np.int64)
    col = ctx.empty(max(cap_entries, 1), np.uint32)
    val = ctx.empty(max(cap_entries, 1), np.uint32)
    nnz = _i64(0)
    ctx.call("skm_dense_to_csr", _i64(n), _i64(ncols), code, _ptr(dense), _i64(ld), _i64(cap_entries), _ptr(rowptr),
             _ptr(col), _ptr(val), C.byref(nnz))
    return CountsCSR(ctx, n, int(nnz.value), 32, rowptr, col, val, None)


def transpose(ctx, n: int, nnz: int, ncols: int, rowptr, colidx, counts):
    """Column-major copy (postings) of a CSR whose column ids are all real (< ncols): a CSR built with
    elide_singletons carries 0xFFFFFFFF markers and must not come here."""
    colptr = ctx.empty(ncols + 1, np.uint32)
    post = ctx.empty(max(nnz, 1), np.uint64)
    ctx.call("skm_csr_transpose", _i64(n), _i64(nnz), _i64(ncols), _ptr(rowptr), _ptr(colidx), _ptr(counts),
             _ptr(colptr), _ptr(post))
    return colptr, post


def cosine_matrix(ctx, x: CountsCSR, x_rnorm, m: int, ncols: int, colptr, post, y_rnorm,
                  row0: int = 0, row1: Optional[int] = None, mode: int = 0, out=None, ld: Optional[int] = None,
                  post_bits: int = 64, postcnt=None):
    """a13/a14: float32 block [row1-row0, m] of cosine similarities (mode 0) or distances (mode 1).
    `post_bits` / `postcnt`: the posting format of `post` (Basis.post_bits / Basis.postcnt)."""
    row1 = x.n if row1 is None else row1
    ld = m if ld is None else ld
    rows = row1 - row0
    if out is None:
        out = ctx.empty((max(rows, 1), max(ld, 1)), np.float32)
    ctx.call(
        "skm_cosine_csr", _i64(x.n), _ptr(x.rowptr), _ptr(x.colidx), _ptr(x.counts), _ptr(x_rnorm), _i64(m), _i64(ncols),
        _ptr(colptr), _ptr(post), post_bits, _ptr(postcnt), _ptr(y_rnorm), _i64(row0), _i64(row1), mode, _ptr(out), _i64(ld),
    )
    return out


def reduced_strings(ctx, batch: SeqBatch, lut: AlphabetLUT, timings: Optional[dict] = None) -> np.ndarray:
    """a3 for a whole batch as the numpy '<U{w}' array the rule stores (`seqs`, rules/kmerize.smk:121-127,136): the
    recoded bytes become UCS-4 rows on the device (skm_rows_to_utf32), no per-record Python string."""
    n = batch.n
    if n == 0:
        return np.array([], dtype=str)
    d_bytes = ctx.empty(batch.total + 64, np.uint8)
    d_len = ctx.empty(n, np.int32)
    ctx.call("skm_recode", _ptr(lut.translate), _ptr(batch.d_seq), _ptr(batch.d_off), _i64(n), _ptr(d_bytes), _ptr(d_len))
    width = max(int(d_len.download(n).max()), 1)  # numpy's width for all-empty strings is 1
    d_out = ctx.empty((n, width), np.uint32)
    ctx.call("skm_rows_to_utf32", _ptr(d_bytes), _ptr(batch.d_off), _ptr(d_len), _i64(n), _i64(width), _ptr(d_out))
    return d_out.download().reshape(n, width).view(f"<U{width}").ravel()


def decode_kmers(ctx, lut: AlphabetLUT, k: int, d_codes, count: int, d_index=None) -> np.ndarray:
    """Codes resident on the device -> numpy '<U{k}' k-mer strings (skm_decode_kmers_utf32); `d_index` (uint32, device)
    selects and orders them."""
    if count == 0:
        return np.array([], dtype=str)
    letters = np.frombuffer(lut.letters.encode("latin-1"), dtype=np.uint8)
    bits = 32 if np.dtype(d_codes.dtype).itemsize == 4 else 64
    d_out = ctx.empty((count, k), np.uint32)
    ctx.call("skm_decode_kmers_utf32", bits, lut.nsym, k, _ptr(letters), _ptr(d_codes), _ptr(d_index), _i64(count), _ptr(d_out))
    return d_out.download().reshape(count, k).view(f"<U{k}").ravel()


def csr_remap_columns(ctx, csr: CountsCSR, d_colmap, ncols: int):
    """(rowptr int64[n+1], col uint32[nnz'], val uint32[nnz']) on the host: the entries of `csr` whose column maps
    through `d_colmap` to a real column, re-labelled, row order kept (skm_csr_remap_columns)."""
    n = csr.n
    cap = max(csr.nnz, 1)
    d_rowptr = ctx.empty(n + 1, np.int64)
    d_col, d_val = ctx.empty(cap, np.uint32), ctx.empty(cap, np.uint32)
    nnz = _i64(0)
    ctx.call("skm_csr_remap_columns", _i64(n), _ptr(csr.rowptr), _ptr(csr.colidx), _ptr(csr.counts), _ptr(d_colmap), _i64(ncols),
             _ptr(d_rowptr), _ptr(d_col), _ptr(d_val), C.byref(nnz))
    nz = int(nnz.value)
    return d_rowptr.download(n + 1), d_col.download(nz), d_val.download(nz)


def csr_to_dense(ctx, n: int, rowptr, colidx, counts, ncols_out: int, colmap=None, presence: bool = False,
                 dtype=np.float64, out=None) -> _hip.DeviceArray:
    code = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int8): 2}[np.dtype(dtype)]
    if out is None or out.shape != (max(n, 1), max(ncols_out, 1)) or out.dtype != np.dtype(dtype):
        out = ctx.empty((max(n, 1), max(ncols_out, 1)), dtype)
    ctx.call("skm_csr_to_dense", _i64(n), _ptr(rowptr), _ptr(colidx), _ptr(counts), _ptr(colmap), _i64(ncols_out),
             1 if presence else 0, code, _ptr(out), _i64(max(ncols_out, 1)))
    return out


def count_dense(ctx, batch: SeqBatch, lut: AlphabetLUT, k: int, dtype=np.uint16, out=None) -> _hip.DeviceArray:
    """North-star dense count matrix [n x |S|^k] by atomic scatter (small bases only)."""
    space = lut.nsym**k
    ld = space + (space & 1)
    if out is None or out.shape != (max(batch.n, 1), ld) or out.dtype != np.dtype(dtype):
        out = ctx.empty((max(batch.n, 1), ld), dtype)
    code = {np.dtype(np.uint16): 0, np.dtype(np.uint32): 1}[np.dtype(dtype)]
    ctx.call("skm_count_dense", _ptr(lut.rank), lut.nsym, k, _ptr(batch.d_seq), _ptr(batch.d_off), _i64(batch.n), code,
             _ptr(out), _i64(ld))
    return out


def matrix_row_stats(ctx, mat, n: int, m: int, ld: int):
    """(float64 row sums, uint32 non-zero counts) of a float32 block resident on the device."""
    d_sum = ctx.empty(max(n, 1), np.float64)
    d_nnz = ctx.empty(max(n, 1), np.uint32)
    ctx.call("skm_matrix_row_stats", _i64(n), _i64(m), _ptr(mat), _i64(ld), _ptr(d_sum), _ptr(d_nnz))
    return d_sum.download(n), d_nnz.download(n)


def csr_max_count(ctx, csr: CountsCSR) -> int:
    mx = C.c_uint32(0)
    ctx.call("skm_csr_max_count", _i64(csr.nnz), _ptr(csr.counts), C.byref(mx))
    return int(mx.value)


def cosine_dense_i8(ctx, n: int, m: int, kdim: int, x_i8, y_i8, x_rnorm, y_rnorm, mode: int = 0, out=None,
                    ld: Optional[int] = None):
    """a13 on the matrix cores: float32 [n x m] from int8 count matrices [n x kdim], [m x kdim]."""
    ld = (m + 3) // 4 * 4 if ld is None else ld
    if out is None:
        out = ctx.empty((max(n, 1), max(ld, 1)), np.float32)
    ctx.call("skm_cosine_dense_i8", _i64(n), _i64(m), _i64(kdim), _ptr(x_i8), _ptr(y_i8), _ptr(x_rnorm), _ptr(y_rnorm),
             mode, _ptr(out), _i64(ld))
    return out


class NeighborLists:
    """Exact sparse Gram rows of a row block: per row the rows j sharing a k-mer and the int32 dot."""

    def __init__(self, ctx, row0, row1, start, length, ent, total, overflow_rows, m=0):
        self.ctx, self.row0, self.row1, self.m = ctx, row0, row1, m  # m: rows of Y (the neighbours' numbering)
        self.start, self.length, self.ent = start, length, ent
        self.total, self.overflow_rows = total, overflow_rows

    def host(self):
        """(start uint64[nrows], len uint32[nrows] (0xFFFFFFFF = not held), j uint32[total], dot int32[total])."""
        nrows = self.row1 - self.row0
        ent = self.ent.download(self.total)
        return (self.start.download(nrows), self.length.download(nrows), (ent >> np.uint64(32)).astype(np.uint32),
                (ent & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32))


def gram_neighbors(ctx, x: CountsCSR, x_rnorm, m: int, ncols: int, colptr, post, y_rnorm, row0: int = 0,
                   row1: Optional[int] = None, cap_entries: Optional[int] = None, post_bits: int = 64,
                   postcnt=None) -> NeighborLists:
    """Neighbour lists (exact sparse Gram rows) for rows [row0,row1) of `x` against the postings of an
    m-row matrix: the reduced output when the dense matrix cannot be stored (skm_gram_neighbors).  `x_rnorm` /
    `y_rnorm` (row_norms of the two matrices) let the device refuse rows whose dot products may not fit the 32-bit
    entries: they come back in `overflow_rows`."""
    row1 = x.n if row1 is None else row1
    nrows = row1 - row0
    cap = int(cap_entries if cap_entries is not None else 16 * max(x.nnz, 1) + (1 << 20))
    start = ctx.empty(max(nrows, 1), np.uint64)
    length = ctx.empty(max(nrows, 1), np.uint32)
    ent = ctx.empty(max(cap, 1), np.uint64)
    total, ovf = _i64(0), _i64(0)
    ctx.call("skm_gram_neighbors", _i64(x.n), _ptr(x.rowptr), _ptr(x.colidx), _ptr(x.counts), _i64(m), _i64(ncols),
             _ptr(colptr), _ptr(post), post_bits, _ptr(postcnt), _ptr(x_rnorm), _ptr(y_rnorm), _i64(row0), _i64(row1), _i64(cap),
             _ptr(start), _ptr(length), _ptr(ent), C.byref(total), C.byref(ovf))
    return NeighborLists(ctx, row0, row1, start, length, ent, int(total.value), int(ovf.value), m=m)


def neighbors_topk(ctx, nb: NeighborLists, x_rnorm, y_rnorm, k: int, exclude_self: bool = True):
    """(idx uint32[nrows,k], score float32[nrows,k]): the k best cosine neighbours of every row."""
    nrows = nb.row1 - nb.row0
    idx = ctx.empty(max(nrows * k, 1), np.uint32)
    val = ctx.empty(max(nrows * k, 1), np.float32)
    ctx.call("skm_neighbors_topk", _i64(nrows), _i64(nb.row0), _ptr(nb.start), _ptr(nb.length), _ptr(nb.ent),
             _ptr(x_rnorm), _ptr(y_rnorm), _i64(getattr(nb, "m", 0)), k, 1 if exclude_self else 0, _ptr(idx), _ptr(val))
    return idx.download(nrows * k).reshape(nrows, k), val.download(nrows * k).reshape(nrows, k)


class DensePipeline:
    """Small-basis variant of Pipeline: the full |S|^k basis as dense int8 counts, cosine by MFMA.

    Column id == k-mer code, so no basis sort is needed.  Raises if a count exceeds 127 (the
    int8 operand range); use Pipeline (exact for any count) in that case."""

    def __init__(self, ctx: _hip.Context, lut: AlphabetLUT, k: int):
        space = lut.nsym**k
        if space > 2**26 or lut.code_bits(k) != 32:
            raise ValueError(f"dense path needs |S|^k <= 2^26 (got {lut.nsym}^{k})")
        self.ctx, self.lut, self.k = ctx, lut, k
        self.space = space
        self.kdim = (space + 127) // 128 * 128  # K-step of the LDS-DMA MFMA kernel
        self.csr = self.rnorm = self.dense = self.out = self._d_max = None

    def step(self, batch: SeqBatch, mode: int = 0):
        """count -> int8 operand -> norms -> symmetric MFMA GEMM.  Every buffer is reused between steps.  The largest
        count (which decides whether int8 holds the operand) stays on the device and is read once, after the GEMM has
        been queued; one size read-back remains in front of it (count_csr returns the entry count to the host)."""
        ctx = self.ctx
        self.csr = count_csr(ctx, batch, self.lut, self.k, out=self.csr)
        n = self.csr.n
        if self._d_max is None:
            self._d_max = ctx.zeros(1, np.uint32)
        ctx.call("skm_csr_max_count_dev", _i64(n), _i64(self.csr.nnz), _ptr(self.csr.rowptr), _ptr(self.csr.counts), _ptr(self._d_max))
        self.rnorm = row_norms(ctx, n, self.csr.rowptr, self.csr.counts, out=self.rnorm)
        # codes double as column ids of the full basis
        self.dense = csr_to_dense(ctx, n, self.csr.rowptr, self.csr.codes, self.csr.counts, self.kdim, dtype=np.int8, out=self.dense)
        ld = (n + 3) // 4 * 4
        if self.out is None or self.out.shape != (max(n, 1), max(ld, 1)):
            self.out = None
            self.out = ctx.empty((max(n, 1), max(ld, 1)), np.float32)
        cosine_dense_i8(ctx, n, n, self.kdim, self.dense, self.dense, self.rnorm, self.rnorm, mode=mode, out=self.out, ld=ld)
        if int(self._d_max.download(1)[0]) > 127:
            self.out = self.dense = None  # they hold a product of saturated operands: nothing a caller may read
            raise OverflowError("a k-mer count exceeds 127: int8 dense path not applicable")
        return self.out


DENSE_ROUTE_MAX_COLS = 1 << 17   # |S|^k up to which the full basis may be held dense (int8, n x |S|^k bytes)
DENSE_ROUTE_MIN_FILL = 0.005     # windows per sequence / |S|^k: below this the sparse kernels do less work


class Pipeline:
    """vectorize + all-pairs cosine for one batch, reusing every device buffer between steps.

    This is the unit `bench.py` times: inputs resident in HBM, outputs (CSR counts, basis,
    N x N float32 cosine) resident in HBM.

    `step` picks between two exact routes from what the host knows before anything runs.  Large or sparsely filled
    bases (BASELINE's red6 k=12: 2e9 possible columns, ~126 observed per sequence) take the sparse route: observed
    basis by a global sort, postings, sparse Gram, streaming writer.  Small full bases (|S|^k <= 2^17 with at least
    0.5 % of a row filled: the reference's CI configuration solvacc k=8 = 6561 columns, hydro k <= 17) take the DENSE
    route: the count stage's CSR becomes an int8 operand whose column ids are the codes themselves (no sort, no
    postings) and the cosine is one symmetric GEMM on the matrix cores (skm_cosine_dense_i8); rows with a count above
    127 are recomputed exactly by skm_cosine_fixup_rows, so the route needs no data-dependent decision on the host.
    `dense_route=False` (or SKM_COSINE_PATH=lists / cursor, the test knobs that name a sparse kernel) keeps the sparse
    route.  `basis` is built on first use after a dense-route step.
    """

    def __init__(self, ctx: _hip.Context, lut: AlphabetLUT, k: int, post32: bool = False, fused: bool = True,
                 dense_route="auto", graphs=False):
        self.ctx, self.lut, self.k = ctx, lut, k
        self.fused = fused  # skm_vectorize_csr (one call, no result read back) instead of the three-call form
        # 4-byte posting words (batches under 2^24 sequences): half the posting bytes, but measured SLOWER
        # end to end on MI355X (k_gram_sparse is bound by instruction issue and the decode costs
        # instructions: 1.90 vs 1.75 ms at BASELINE configs[2]), so it is opt-in
        self.post32 = post32
        self.dense_route = dense_route
        self.route = "sparse"      # route of the last step
        self.csr: Optional[CountsCSR] = None
        self._basis: Optional[Basis] = None
        self._basis_stale = False  # a dense-route step builds no basis; `basis` does, on first use
        self.rnorm = None
        self.out = None
        self._dense = self._irr_list = self._irr_count = None
        self._dense_valid = False
        # HIP-graph replay of whole steps (skm_graph_*): "auto" = batches of at most GRAPH_AUTO_RESIDUES residues, whose step is
        # a few dozen launches of microseconds each (the reference's job size: one FASTA file per job,
        # snekmer/rules/kmerize.smk:57-65); True = any size; False (default) = never.  Opt-in because it buys nothing on
        # the host it was measured on: a 1 000-sequence step is 24 dispatches of >= 5 us each ON THE DEVICE (0.164 ms eager,
        # 0.172 ms replayed); the host's launches were already hidden behind them.  It pays on a slower or busier host.  See `step`.
        self.graphs = graphs
        self._graphs = {}
        self.graph_replays = 0

    @property
    def basis(self) -> Optional[Basis]:
        if self._basis_stale:
            self._basis = build_basis(self.ctx, self.csr, self.lut.nsym, self.k, elide_singletons=True, post32=self.post32)
            self._basis_stale = False
        return self._basis

    @basis.setter
    def basis(self, value):
        self._basis, self._basis_stale = value, False

    def _sparse_forced(self) -> bool:
        return self.dense_route in (False, "never") or _hip.get_option("SKM_COSINE_PATH") in ("lists", "cursor")

    def wants_dense(self, batch: SeqBatch) -> bool:
        """The host-side routing rule: a function of (alphabet, k, batch shape) only."""
        if self._sparse_forced() or batch.n < 1 or batch.total < 1 or self.lut.code_bits(self.k) != 32:
            return False
        space = self.lut.nsym**self.k
        if self.dense_route == "always":
            return space <= (1 << 26)
        fill = max(batch.total / batch.n - self.k + 1, 0.0) / space
        return space <= DENSE_ROUTE_MAX_COLS and fill >= DENSE_ROUTE_MIN_FILL

    def vectorize(self, batch: SeqBatch) -> CountsCSR:
        self.route, self._dense_valid = "sparse", False
        if self.fused and not self.post32 and batch.n >= 1 and batch.total >= 1:
            # one call, no size read back: within a step the host runs ahead of the GPU (one wait per call remains,
            # for the size-class histogram: see vectorize_fused)
            self.csr, self.basis, self.rnorm = vectorize_fused(self.ctx, batch, self.lut, self.k, csr=self.csr,
                                                               basis=self._basis, rnorm=self.rnorm)
            return self.csr
        self.csr = count_csr(self.ctx, batch, self.lut, self.k, out=self.csr)
        self.basis = build_basis(self.ctx, self.csr, self.lut.nsym, self.k, out=self._basis, elide_singletons=True,
                                 post32=self.post32)
        self.rnorm = row_norms(self.ctx, self.csr.n, self.csr.rowptr, self.csr.counts, out=self.rnorm)
        return self.csr

    def _out_block(self, rows: int, ld: int):
        if self.out is None or self.out.shape[0] < rows or self.out.shape[1] != ld:
            self.out = None
            self.out = self.ctx.empty((max(rows, 1), max(ld, 1)), np.float32)
        return self.out

    def cosine(self, row0: int = 0, row1: Optional[int] = None):
        n = self.csr.n
        row1 = n if row1 is None else row1
        if not 0 <= row0 <= row1 <= n:
            raise ValueError(f"bad row range [{row0},{row1}) of {n}")
        ld = (n + 3) // 4 * 4
        if self.route == "dense" and not self._sparse_forced():
            return self._cosine_dense(row0, row1, ld)
        out = self._out_block(row1 - row0, ld)
        b = self.basis
        cosine_matrix(self.ctx, self.csr, self.rnorm, n, b.ncols_hint(), b.colptr, b.post, self.rnorm,
                      row0=row0, row1=row1, out=out, ld=ld, post_bits=b.post_bits, postcnt=b.postcnt)
        return out

    def _cosine_dense(self, row0: int, row1: int, ld: int):
        ctx, n = self.ctx, self.csr.n
        kdim = (self.lut.nsym**self.k + 127) // 128 * 128
        if not self._dense_valid:
            if self._dense is None or self._dense.shape != (n, kdim):
                self._dense = None
                self._dense = ctx.empty((n, kdim), np.int8)
            if self._irr_list is None or self._irr_list.size < n:
                self._irr_list = ctx.empty(n, np.uint32)
            if self._irr_count is None:
                self._irr_count = ctx.zeros(1, np.uint32)
            ctx.call("skm_csr_to_dense_i8", _i64(n), _ptr(self.csr.rowptr), _ptr(self.csr.codes), _ptr(self.csr.counts), _i64(kdim),
                     _ptr(self._dense), _ptr(self._irr_list), _ptr(self._irr_count))
            self._dense_valid = True
        rows = row1 - row0
        out = self._out_block(rows, ld)
        if rows > 0:
            whole = row0 == 0 and row1 == n  # X is Y: the symmetric launch
            cosine_dense_i8(ctx, rows, n, kdim, self._dense if whole else self._dense.at(row0 * kdim), self._dense,
                            self.rnorm if whole else self.rnorm.at(row0), self.rnorm, out=out, ld=ld)
            ctx.call("skm_cosine_fixup_rows", _i64(n), _ptr(self.csr.rowptr), _ptr(self.csr.codes), _ptr(self.csr.counts),
                     _ptr(self.rnorm), _i64(row0), _i64(row1), _ptr(self._irr_list), _ptr(self._irr_count), 0, _ptr(out), _i64(ld))
        return out

    def irregular_rows(self) -> int:
        """Rows the last dense-route step recomputed exactly (a count above 127); reads one word back."""
        return int(self._irr_count.download(1)[0]) if self._dense_valid else 0

    def _step_eager(self, batch: SeqBatch):
        if self.wants_dense(batch):
            self.csr, self.rnorm = vectorize_counts(self.ctx, batch, self.lut, self.k, csr=self.csr, rnorm=self.rnorm)
            self.csr.colidx = None
            self.route, self._basis_stale, self._dense_valid = "dense", True, False
            return self.cosine()
        self.vectorize(batch)
        return self.cosine()

    # ---- whole steps as HIP graphs
    GRAPH_AUTO_RESIDUES = 1 << 23  # "auto": up to ~28 k sequences of 300 aa; above, the launches are a percent of the step
    GRAPH_ENV = _hip.OPTION_NAMES  # host-side switches between exact kernels: a capture freezes them, so they are part of its key
    GRAPH_SLOTS = 8

    def _graph_key(self, batch: SeqBatch):
        if self.graphs is False or getattr(self.ctx, "profiling", False) or batch.ctx is not self.ctx or _hip.under_profiler():
            return None
        if not self.fused or self.post32 or batch.n < 1 or batch.total < 1 or batch.max_len < 1:
            return None  # (those forms wait for the device inside the step)
        if self.graphs == "auto" and batch.total > self.GRAPH_AUTO_RESIDUES:
            return None
        return (batch.d_seq.ptr, batch.d_seq.nbytes, batch.d_off.ptr, batch.n, batch.total, batch.max_len,
                tuple(_hip.get_option(v) for v in self.GRAPH_ENV), self.dense_route)

    def _signature(self):
        """(address, bytes) of every buffer the kernels of the last step touched besides the batch and the context's scratch:
        a captured step may be replayed only while these are what the pipeline still holds."""
        csr = self.csr
        if csr is None or self.out is None or self.rnorm is None:
            return None
        bufs = [csr.rowptr, csr.codes, csr.counts, self.rnorm, self.out]
        if self.route == "dense":
            bufs += [self._dense, self._irr_list, self._irr_count]
        else:
            b = self._basis
            if b is None:
                return None
            bufs += [csr.colidx, b.codes, b.colptr, b.post, b.d_ncols]
        if any(x is None for x in bufs):
            return None
        return (self.route,) + tuple((x.ptr, x.nbytes) for x in bufs)

    def step(self, batch: SeqBatch):
        """vectorize + cosine of one batch.  A step whose batch (same device buffers, same shape) has been seen before is
        replayed as ONE HIP graph: the first sight runs eagerly (it also sizes every buffer and the context's scratch), the
        second is captured (skm_graph_begin .. skm_graph_end record the same library calls instead of running them) and
        launched, later ones are one launch each.  A replay runs the same kernels on the same buffers and reads whatever the
        batch's buffers hold now.  It is only used while the pipeline still holds exactly the buffers the capture saw
        (`_signature`), the library's scratch has not moved (SKM_E_STALE) and per-kernel timing is off; anything else runs
        eagerly.  Results are the eager step's bit for bit (tests/test_gpu_parity.py)."""
        key = self._graph_key(batch)
        if key is None:
            return self._step_eager(batch)
        ent = self._graphs.get(key)
        if ent is not None and ent["graph"] is not None and ent["sig"] is not None and ent["sig"] == self._signature_for(ent):
            try:
                ent["graph"].launch()
            except _hip.HipError as err:
                if err.code != -7:  # not SKM_E_STALE
                    raise
                ent["graph"].close()
                ent["graph"] = None
            else:
                self._restore(ent)
                self.graph_replays += 1
                return self.out
        if ent is None:
            out = self._step_eager(batch)
            while len(self._graphs) >= self.GRAPH_SLOTS:
                old = self._graphs.pop(next(iter(self._graphs)))
                if old["graph"] is not None:
                    old["graph"].close()
            self._graphs[key] = {"graph": None, "sig": self._signature(), "state": self._state()}
            return out
        if ent["graph"] is None and ent["sig"] is not None and ent["sig"] == self._signature_for(ent):
            # second sight, nothing reallocated since the eager run: record the same calls, then launch the recording
            self.ctx.graph_begin()
            try:
                out = self._step_eager(batch)
            except Exception as err:
                try:
                    self.ctx.graph_end().close()
                except _hip.HipError:
                    pass
                if isinstance(err, _hip.HipError) and err.code in (-5, -7):
                    # a scratch slot wanted to grow inside the capture (e.g. the heavy-row panels, switched on by the previous
                    # step's hint, size their blocks for the first time now): nothing was executed - the calls were being
                    # recorded -, so run the step for real, outside a capture, and start over with this shape
                    out = self._step_eager(batch)
                    ent.update(graph=None, sig=self._signature(), state=self._state())
                    return out
                del self._graphs[key]
                raise
            graph = self.ctx.graph_end()
            if self._signature() != ent["sig"]:  # a buffer was replaced inside the capture: the recording is of the new ones
                ent["sig"] = self._signature()
            ent["graph"], ent["state"] = graph, self._state()
            graph.launch()
            self.graph_replays += 1
            return out
        out = self._step_eager(batch)  # buffers changed since this shape was last seen: start over
        if ent["graph"] is not None:
            ent["graph"].close()
        ent.update(graph=None, sig=self._signature(), state=self._state())
        return out

    def _state(self):
        return (self.route, self._basis_stale, self._dense_valid, self.csr.n, self.csr.elided)

    def _signature_for(self, ent):
        """The signature the pipeline's CURRENT buffers would give for the route the entry's step took."""
        route, self.route = self.route, ent["state"][0]
        try:
            return self._signature()
        finally:
            self.route = route

    def _restore(self, ent):
        """Host-side state after a replay: what `_step_eager` leaves behind (sizes that live on the device are fetched lazily)."""
        self.route, self._basis_stale, self._dense_valid, self.csr.n, self.csr.elided = ent["state"]
        self.csr.nnz = None
        if self.route == "dense":
            self.csr.colidx = None
        elif self._basis is not None:
            self._basis.ncols = None

    def drop_graphs(self):
        for ent in self._graphs.values():
            if ent["graph"] is not None:
                ent["graph"].close()
        self._graphs.clear()


class OverlappedPipeline:
    """Pipeline for a STREAM of batches: while batch i's cosine runs on the main context, batch i+1 is vectorized on a
    second context (its own HIP stream and scratch) into a second set of buffers.  The stages in front of the writer
    are latency- and issue-bound (0.13-0.29 of HBM each) and the writer is store-bound; side by side on
    the whole chip each crawls in the other's wave slots (10.5 against 10.8 ms per step), so the side context's stream is
    confined to half of the compute units (skm_create_confined): 9.66 ms per step.  The main stream then carries the
    sparse Gram and the writer, the side stream less work than that: with `side_list_fraction` > 0 the neighbour lists
    of the LAST rows of a batch (that fraction of them) are built on the side context too, right behind its vectorize
    (skm_cosine_csr_phase 1), and the main stream only runs the writer for those rows (phase 2, reading the side
    context's scratch).  The lists live in a context's scratch, so two side contexts alternate.  Results are the
    single-stream Pipeline's, bit for bit (same kernels, same inputs; tests/test_gpu_parity.py).

        pipe = OverlappedPipeline(ctx, lut, k)
        pipe.prefetch(batch0)
        for nxt in batches[1:] + [None]:
            out = pipe.step(nxt)       # cosine of the prefetched batch; vectorize of `nxt` starts beside it

    `out` (float32 [n, ld], HBM) is shared by all steps: consume it (or copy it) before the next step's writer runs,
    i.e. before calling step() again, exactly as with Pipeline.  The streams must sit on different hardware queues:
    snekmer_amd._hip asks the HIP runtime for eight (GPU_MAX_HW_QUEUES) when it loads the library first."""

    EV_VEC, EV_COS = 0, 4  # event slots: EV_VEC + set on the side contexts, EV_COS + set on the main one
    SIDE_CU_GROUPS = (0, 3)  # measured on the bench workload: groups 0-0 12.7 ms, 0-1 10.1, 0-2 10.0, 0-3 9.66, 0-4 10.35, all 10.5
    SIDE_LIST_FRACTION = 0.6  # bench workload, ms per step: 0 9.75, 0.2 9.74, 0.4 9.47, 0.5 9.29-9.36, 0.6 9.25, 0.7 9.27, 0.8 9.63, 1 9.99
    SPLIT_MIN_ROWS = 16384  # below this a batch's lists all stay on the main context
    DEPTH = 1  # batches in flight on side contexts (each on its own context and buffer set)

    def __init__(self, ctx: _hip.Context, lut: AlphabetLUT, k: int, side_ctx: Optional[_hip.Context] = None,
                 side_list_fraction: Optional[float] = None, depth: Optional[int] = None):
        self.ctx, self.lut, self.k = ctx, lut, k
        self.fraction = self.SIDE_LIST_FRACTION if side_list_fraction is None else float(side_list_fraction)
        self.depth = self.DEPTH if depth is None else int(depth)
        if not 1 <= self.depth <= 3:
            raise ValueError("depth: 1, 2 or 3 batches ahead")

        def confined():
            try:  # half of the compute units: what the next batch's kernels may fill beside this batch's writer
                return _hip.Context(ctx.device, cu_groups=self.SIDE_CU_GROUPS)
            except _hip.HipError:  # a device the CU-group masks are not defined for
                return _hip.Context(ctx.device)

        first = side_ctx if side_ctx is not None else confined()
        nsets = self.depth + 1
        self.sides = [first] * nsets
        if self.fraction > 0 or self.depth > 1:  # lists on the side (they live in a context's scratch), or batches side by side: one side context per buffer set
            self.sides = [first] + [_hip.Context(ctx.device, cu_groups=first.cu_groups) if first.cu_groups else _hip.Context(ctx.device)
                                    for _ in range(nsets - 1)]
        self.side = first
        self._owned = [c for c in dict.fromkeys(self.sides) if c is not side_ctx]
        self.sets = [[None, None, None] for _ in range(nsets)]  # (csr, basis, rnorm) per buffer set
        self.queue = []     # sets holding a vectorized batch that has not been consumed yet, oldest first
        self.nxt = 0
        self.out = None
        self.csr = self.basis = self.rnorm = None  # the set the last step() consumed

    @property
    def ready(self):
        """The set the next step() consumes (None: nothing prefetched)."""
        return self.queue[0] if self.queue else None

    def _split_row(self, n: int) -> int:
        """Rows [0, r) get their lists on the main context, rows [r, n) on the side context."""
        if self.fraction <= 0 or n < self.SPLIT_MIN_ROWS:
            return n
        return max(0, min(n, int(n * (1.0 - self.fraction)) // 8 * 8))

    def _block(self, call_ctx, s: int, row0: int, row1: int, phase: int, out_ptr: int, ld: int):
        csr, b, rnorm = self.sets[s]
        args = (_i64(csr.n), _ptr(csr.rowptr), _ptr(csr.colidx), _ptr(csr.counts), _ptr(rnorm), _i64(csr.n), _i64(b.ncols_hint()),
                _ptr(b.colptr), _ptr(b.post), b.post_bits, _ptr(b.postcnt), _ptr(rnorm), _i64(row0), _i64(row1), 0,
                C.c_void_p(out_ptr) if out_ptr else None, _i64(ld))
        if phase == 0:
            call_ctx.call("skm_cosine_csr", *args)
        else:
            call_ctx.call("skm_cosine_csr_phase", self.ctx.handle if phase == 2 else None, phase, *args)

    def prefetch(self, batch: SeqBatch) -> None:
        """Vectorize `batch` on a side context into the free buffer set (and build the lists of its last rows there)."""
        if len(self.queue) >= self.depth:
            raise RuntimeError("a prefetched batch is waiting: call step() first")
        s = self.nxt
        side = self.sides[s]
        # the set's previous contents (and that side context's lists) were last read by the cosine depth + 1 steps ago
        side.wait_event(self.ctx, self.EV_COS + s)
        if batch.ctx is not self.ctx and all(batch.ctx is not c for c in self.sides):
            raise ValueError("the batch must live on the pipeline's device")
        if getattr(batch, "ready", None) is not None:  # an asynchronous upload (BatchUploader): ordered on the device
            side.wait_event(*batch.ready)
        csr, basis, rnorm = self.sets[s]
        self.sets[s] = list(vectorize_fused(side, batch, self.lut, self.k, csr=csr, basis=basis, rnorm=rnorm))
        n = self.sets[s][0].n
        r = self._split_row(n)
        if r < n:
            self._block(side, s, r, n, 1, 0, (n + 3) // 4 * 4)
        side.record_event(self.EV_VEC + s)
        if getattr(batch, "on_consumed", None) is not None:  # the batch's buffers may be refilled behind this event
            batch.on_consumed(side, self.EV_VEC + s)
        self.queue.append(s)
        self.nxt = (s + 1) % len(self.sets)

    def step(self, next_batch: Optional[SeqBatch] = None):
        """Cosine of the prefetched batch (queued on the main context), then the prefetch of `next_batch`."""
        if not self.queue:
            raise RuntimeError("nothing prefetched: call prefetch(batch) first")
        s = self.queue.pop(0)
        self.csr, self.basis, self.rnorm = self.sets[s]
        self.ctx.wait_event(self.sides[s], self.EV_VEC + s)
        n = self.csr.n
        ld = (n + 3) // 4 * 4
        if self.out is None or self.out.shape != (max(n, 1), max(ld, 1)):
            self.out = None
            self.out = self.ctx.empty((max(n, 1), max(ld, 1)), np.float32)
        r = self._split_row(n)
        # (Round 6 measured the lists of rows [0, r) on a THIRD context, so that the main stream carries the two writer
        # launches only: 10.6 / 11.4 / 12.0 ms per step with that context on CU groups 4-7 / 0-3 / 4-5 against 9.45 without
        # it - a third latency-bound stream slows the writer and the side stream more than it hides: not kept.)
        if r > 0 or n == 0:
            self._block(self.ctx, s, 0, r, 0, self.out.ptr, ld)
        if r < n:
            self._block(self.sides[s], s, r, n, 2, self.out.at(r * ld), ld)
        self.ctx.record_event(self.EV_COS + s)
        if next_batch is not None:
            self.prefetch(next_batch)
        return self.out

    def contexts(self):
        """Every context besides the main one that carries work of this pipeline (for per-kernel timing, waiting, closing)."""
        return list(dict.fromkeys(self.sides))

    def sync(self):
        for c in self.contexts():
            c.sync()
        self.ctx.sync()

    def close(self):
        """Wait for everything queued, then close the contexts this pipeline created (their streams go back to the
        library's cache)."""
        self.sync()
        for c in self._owned:
            c.close()
        self._owned = []
