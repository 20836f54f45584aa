"""dist: sequence-sharded vectorize + row-block cosine across the GPUs of one node.

The reference has no distributed layer; its only parallelism is one Snakemake job per FASTA
file (snekmer/rules/kmerize.smk:57-65).  Here sequences shard across ranks (one process per
GPU).  A k-mer's integer code is a pure function of (alphabet, k, window), so shards agree on
column identity without exchanging a dictionary; the single exchange is an all-gather of the
per-shard CSR rows, after which rank r computes rows [lo_r, hi_r) of the
similarity matrix against all N columns.

Two forms of the exchange (`ShardedPipeline(basis=...)`):

* ``"distributed"`` (default): the postings are built in parallel.  Every rank groups its entries
  by OWNER rank (a fixed hash of the k-mer code), one all-to-all moves them, each owner sorts its
  share (1/G of the entries) and builds the postings of its k-mers, and one all-gather of the
  postings, the owners' column tables and the row norms gives every rank the full column-major
  copy; a rank then finds the columns of its own rows with one probe of the owner's hash table.
  No rank ever sorts the whole matrix.  Four collectives per step: two small size exchanges, one
  grouped all-to-all (codes + posting words), one grouped all-gather (five arrays).
* ``"replicated"``: all-gather of the raw CSR shards, then every rank builds the basis of the full
  matrix itself (simple; the replicated sort is the Amdahl term of strong scaling).

`plan_*` functions are pure host logic (covered by world_size-2 gloo tests on CPU);
`RcclExchange` is the device implementation over skm_allgatherv / skm_alltoallv.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

_p = C.c_void_p
_i64 = C.c_int64


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal row blocks: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_bounds_by_residues(offsets: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks balanced by residue count (stage 1 cost is per residue)."""
    n = len(offsets) - 1
    total = int(offsets[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(offsets, total * r / world, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.minimum(cuts, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def plan_allgather(nnz_per_rank: Sequence[int], rows_per_rank: Sequence[int], code_bytes: int, count_bytes: int = 4):
    """Byte counts of the three all-gathers (codes, counts, local rowptr) per rank.  Counts travel
    as single bytes (count_bytes=1) when every rank's largest count is <= 255, which is the rule
    for k-mer counts; 4 otherwise."""
    return {
        "codes": [int(z) * code_bytes for z in nnz_per_rank],
        "counts": [int(z) * count_bytes for z in nnz_per_rank],
        "rowptr": [(int(r) + 1) * 8 for r in rows_per_rank],
    }


def plan_alltoall(counts: np.ndarray, rank: int, itemsize: int):
    """Byte sizes of one all-to-all from the gathered [src, dst] matrix of entry counts:
    (send_bytes[dst], recv_bytes[src]) for `rank`."""
    counts = np.asarray(counts, dtype=np.int64)
    return counts[rank, :] * itemsize, counts[:, rank] * itemsize


def plan_alltoallv_host(elem_bytes: Sequence[int], send_counts: Sequence[int], recv_counts: Sequence[int]):
    """Host statement of skm_plan_alltoallv: ops[p][a] = (send_off, send_bytes, recv_off, recv_bytes) of what this
    rank exchanges with peer p for array a; segments back to back in rank order in every buffer."""
    world = len(send_counts)
    ops = [[None] * len(elem_bytes) for _ in range(world)]
    for a, eb in enumerate(elem_bytes):
        so = ro = 0
        for p in range(world):
            sb, rb = int(send_counts[p]) * int(eb), int(recv_counts[p]) * int(eb)
            ops[p][a] = (so, sb, ro, rb)
            so += sb
            ro += rb
    return ops


def plan_allgatherv_host(rank: int, elem_bytes: Sequence[int], counts):
    """Host statement of skm_plan_allgatherv: counts[a][r] elements of array a come from rank r; every peer gets
    this rank's whole contribution (send_off 0)."""
    counts = np.asarray(counts, dtype=np.int64)
    world = counts.shape[1]
    ops = [[None] * len(elem_bytes) for _ in range(world)]
    for a, eb in enumerate(elem_bytes):
        ro = 0
        for p in range(world):
            rb = int(counts[a, p]) * int(eb)
            ops[p][a] = (0, int(counts[a, rank]) * int(eb), ro, rb)
            ro += rb
    return ops


def owner_host(codes: np.ndarray, world: int) -> np.ndarray:
    """Host statement of the device's bucket_of (snekmer_amd/csrc/skm_shard.hip): owner rank of
    every k-mer code.  uint32 codes and uint64 codes hash differently, as on the device."""
    codes = np.asarray(codes)
    if codes.dtype == np.uint64:
        m = codes * np.uint64(0x9E3779B97F4A7C15)
        x = ((m >> np.uint64(32)) ^ (m & np.uint64(0xFFFFFFFF))).astype(np.uint64)
    else:
        x = codes.astype(np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    h = (x * np.uint64(0x9E3779B1)) & mask
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & mask
    return ((h * np.uint64(world)) >> np.uint64(32)).astype(np.int64)


def postings_host(codes: np.ndarray, rowcount: np.ndarray):
    """Host statement of skm_bucket_postings: from the (code, row | count << 32) entries an owner
    received -> (distinct k-mers, codes of its shared columns ascending, start of every shared
    column in `post`, post)."""
    order = np.argsort(codes, kind="stable")
    sk, rc = codes[order], rowcount[order]
    head = np.r_[True, sk[1:] != sk[:-1]] if len(sk) else np.zeros(0, bool)
    more = np.r_[sk[1:] == sk[:-1], False] if len(sk) else np.zeros(0, bool)
    shared = ~head | more
    post = rc[shared]
    col_head = head & more
    starts = np.cumsum(shared)[col_head] - 1
    return int(head.sum()), sk[col_head], starts.astype(np.uint32), post


def owner_answers_host(codes: np.ndarray) -> np.ndarray:
    """Host statement of skm_bucket_postings' d_ret: for every entry an owner received (in the order received) the
    owner-local column of its k-mer - columns are the owner's k-mers found in >= 2 rows, in ascending code order -,
    0xFFFFFFFF for a k-mer of one row."""
    codes = np.asarray(codes)
    uniq, inv, cnt = np.unique(codes, return_inverse=True, return_counts=True)
    col_of_uniq = np.where(cnt >= 2, np.cumsum(cnt >= 2) - 1, 0xFFFFFFFF).astype(np.uint32)
    return col_of_uniq[inv] if len(codes) else np.zeros(0, np.uint32)


def concat_rowptr_host(local_rowptrs: Sequence[np.ndarray]) -> np.ndarray:
    """Host statement of skm_csr_concat_rowptr (used by the CPU tests to check the plan)."""
    out, base = [], 0
    for rp in local_rowptrs:
        out.append(np.asarray(rp[:-1], dtype=np.int64) + base)
        base += int(rp[-1])
    out.append(np.asarray([base], dtype=np.int64))
    return np.concatenate(out)


class RcclExchange:
    """All-gather of CSR shards over RCCL (skm_comm_* / skm_allgatherv)."""

    def __init__(self, ctx, world: int, rank: int, unique_id: bytes):
        self.ctx, self.world, self.rank = ctx, world, rank
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        ctx.call("skm_comm_init", world, rank, buf.ctypes.data_as(_p))
        self._small = {}  # (send, recv) device buffers of allgather_i64, by vector length

    @staticmethod
    def new_unique_id() -> bytes:
        from . import _hip

        lib = _hip.load_library()
        buf = np.zeros(_hip.COMM_ID_BYTES, dtype=np.uint8)
        _hip._check(lib, lib.skm_comm_unique_id(buf.ctypes.data_as(_p)))
        return buf.tobytes()

    def allgather_i64(self, values) -> np.ndarray:
        """Gather a small fixed-length int64 vector from every rank -> [world, len]."""
        vals = np.atleast_1d(np.asarray(values, dtype=np.int64))
        if vals.size not in self._small:
            self._small[vals.size] = (self.ctx.empty(vals.size, np.int64), self.ctx.empty(self.world * vals.size, np.int64))
        mine, gathered = self._small[vals.size]
        mine.upload(vals)
        sizes = np.full(self.world, 8 * vals.size, dtype=np.int64)
        self.ctx.call("skm_allgatherv", _p(mine.ptr), sizes.ctypes.data_as(_p), _p(gathered.ptr))
        return gathered.download(self.world * vals.size).reshape(self.world, vals.size)

    def allgather_i64_dev(self, d_vals, count: int) -> np.ndarray:
        """The same for `count` int64 values that already sit on the device (sizes a kernel just produced): gathered
        device to device, then ONE copy to the host -> [world, count].  One host round trip per size exchange."""
        key = ("dev", count)
        if key not in self._small:
            self._small[key] = self.ctx.empty(self.world * count, np.int64)
        gathered = self._small[key]
        sizes = np.full(self.world, 8 * count, dtype=np.int64)
        self.ctx.call("skm_allgatherv", _p(d_vals.ptr), sizes.ctypes.data_as(_p), _p(gathered.ptr))
        return gathered.download(self.world * count).reshape(self.world, count)

    def allgatherv(self, d_send, nbytes_per_rank: Sequence[int], d_recv):
        sizes = np.asarray(nbytes_per_rank, dtype=np.int64)
        self.ctx.call("skm_allgatherv", _p(d_send.ptr), sizes.ctypes.data_as(_p), _p(d_recv.ptr))

    def alltoallv(self, d_send, send_bytes: Sequence[int], d_recv, recv_bytes: Sequence[int]):
        sb = np.ascontiguousarray(send_bytes, dtype=np.int64)
        rb = np.ascontiguousarray(recv_bytes, dtype=np.int64)
        self.ctx.call("skm_alltoallv", _p(d_send.ptr), sb.ctypes.data_as(_p), _p(d_recv.ptr), rb.ctypes.data_as(_p))

    @staticmethod
    def _ptr_array(arrays):
        return (C.c_void_p * len(arrays))(*[a.ptr for a in arrays])

    def alltoallv_multi(self, sends, recvs, elem_bytes: Sequence[int], send_counts: Sequence[int], recv_counts: Sequence[int]):
        """Several parallel arrays in one grouped all-to-all (skm_alltoallv_multi): send_counts[p] elements of
        every array go to rank p, recv_counts[p] arrive from it."""
        eb = np.ascontiguousarray(elem_bytes, dtype=np.int64)
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        self.ctx.call("skm_alltoallv_multi", len(sends), self._ptr_array(sends), self._ptr_array(recvs), eb.ctypes.data_as(_p),
                      sc.ctypes.data_as(_p), rc.ctypes.data_as(_p))

    def allgatherv_multi(self, sends, recvs, elem_bytes: Sequence[int], counts):
        """Several arrays in one grouped all-gather (skm_allgatherv_multi): counts[a][r] elements of array a come
        from rank r; every contribution lands at its final offset of recvs[a]."""
        eb = np.ascontiguousarray(elem_bytes, dtype=np.int64)
        cn = np.ascontiguousarray(counts, dtype=np.int64)
        self.ctx.call("skm_allgatherv_multi", len(sends), self._ptr_array(sends), self._ptr_array(recvs), eb.ctypes.data_as(_p),
                      cn.ctypes.data_as(_p))


class _ColumnMajor:
    """Postings of the full matrix as the cosine kernels take them (colptr/post), plus the number of
    distinct k-mers (`ncols`, single-row ones included) for reporting."""

    def __init__(self):
        self.ncols = 0          # distinct k-mers of the whole batch
        self.ncols_shared = 0   # those found in >= 2 rows: the columns of colptr/post
        self.colptr = self.post = None


class ShardedPipeline:
    """vectorize the local shard, exchange, then cosine (or top-k) for the local row block."""

    MAX_SHARD_RESIDUES = (1 << 30) - 1  # one digit of the one-sweep sort groups a shard by owner: 30-bit positions

    def __init__(self, ctx, lut, k: int, exchange, bounds: Sequence[Tuple[int, int]], total_residues: int,
                 basis: Optional[str] = None, columns: Optional[str] = None):
        import os

        from . import engine

        if basis is None:  # SKM_DIST_BASIS=replicated selects the all-gather-only exchange without a code change
            basis = os.environ.get("SKM_DIST_BASIS", "distributed")
        if basis not in ("distributed", "replicated"):
            raise ValueError("basis must be 'distributed' or 'replicated'")
        self.engine = engine
        self.ctx, self.lut, self.k, self.ex = ctx, lut, k, exchange
        self.mode = basis
        # how a rank learns the column ids of its own entries (distributed basis): "owners" = every owner answers the entries
        # it received with one uint32 each through the reverse of the all-to-all (no hash table is built, gathered or
        # probed); "tables" = round 4's form: owners' hash tables all-gathered, one probe per entry
        self.columns = columns or os.environ.get("SKM_DIST_COLUMNS", "owners")
        if self.columns not in ("owners", "tables"):
            raise ValueError("columns must be 'owners' or 'tables'")
        self.bounds = list(bounds)
        self.rank = exchange.rank
        self.world = len(self.bounds)
        self.n_total = self.bounds[-1][1]
        self.rows = [hi - lo for lo, hi in self.bounds]
        bits = lut.code_bits(k)
        self.code_bits = bits
        self.code_dtype = np.uint32 if bits == 32 else np.uint64
        self.local = None
        self.basis = None
        self.rnorm = None
        self.out = None
        self.nnz_total = 0
        self.sizes = {}
        self.x = None  # the CSR view the cosine kernels read rows [lo, hi) from
        if basis == "replicated":
            cap = total_residues + 1
            self.bytes_local = ctx.empty(cap, np.uint8)   # counts of the local shard, one byte each
            self.bytes_full = ctx.empty(cap, np.uint8)
            self.g_rowptr_local = ctx.empty(self.n_total + len(self.bounds), np.int64)
            self.full = engine.CountsCSR(ctx, self.n_total, 0, bits, ctx.empty(self.n_total + 1, np.int64),
                                         ctx.empty(cap, self.code_dtype), ctx.empty(cap, np.uint32), None)
        else:
            self.full = None
            self._buf = {}
            self.rowptr_g = ctx.empty(self.n_total + 1, np.int64)
            self.rnorm = ctx.empty(self.n_total + 4, np.float32)

    # ------------------------------------------------------------------ buffers (distributed mode)
    def _need(self, name: str, size: int, dtype):
        cur = self._buf.get(name)
        size = max(int(size), 1)
        if cur is None or cur.size < size or cur.dtype != np.dtype(dtype):
            self._buf[name] = None
            cur = self._buf[name] = self.ctx.empty(size + size // 8, dtype)  # head-room: sizes drift between steps
        return cur

    def exchange(self, shard_batch):
        """vectorize the local shard and exchange: afterwards `self.x` (rows [lo, hi) with global row
        numbers and column ids), `self.basis` (colptr/post of the full matrix) and `self.rnorm` (all N
        rows) are set on every rank."""
        if self.mode == "replicated":
            return self._exchange_replicated(shard_batch)
        return self._exchange_distributed(shard_batch)

    def _gather_sizes(self, d_vals, count: int) -> np.ndarray:
        """[world, count] int64 from `count` device-resident values per rank: one host round trip when the exchange can
        gather device buffers (RcclExchange), else a download followed by the exchange's host collective (test doubles)."""
        if hasattr(self.ex, "allgather_i64_dev"):
            return self.ex.allgather_i64_dev(d_vals, count)
        return self.ex.allgather_i64(d_vals.download(count))

    def _exchange_distributed(self, shard_batch):
        """TWO host round trips per step (the [src, dst] entry counts and the owners' sizes, each gathered device to
        device and read back once): count stage, owner grouping and owner-side postings never wait for the device
        (sizes stay in d_rowptr[n] / device counters, launches cover the shard's capacity)."""
        e, ctx, ex, G, me = self.engine, self.ctx, self.ex, self.world, self.rank
        lo, hi = self.bounds[me]
        nloc = hi - lo
        cb = np.dtype(self.code_dtype).itemsize
        rn = self._need("rn_local", nloc + 4, np.float32)
        d_counts = self._need("d_counts", G, np.int64)
        if shard_batch.n >= 1 and shard_batch.total >= 1:
            # counts + norms of the local rows in one call that reads nothing back (skm_vectorize_csr, count stage only)
            self.local, _ = e.vectorize_counts(ctx, shard_batch, self.lut, self.k, csr=self.local, rnorm=rn)
            loc, cap = self.local, shard_batch.total + 1
        else:  # an empty shard (a world larger than the batch)
            self.local = loc = e.count_csr(ctx, shard_batch, self.lut, self.k, out=self.local)
            e.row_norms(ctx, nloc, loc.rowptr, loc.counts, out=rn)
            cap = 1
        # 1. entries grouped by owner rank (sized by the capacity; the entry count stays on the device).  The grouping is one
        # digit of the library's one-sweep sort, whose tile words hold 30-bit positions: a shard is limited to 2^30 - 1
        # residues (3.5 M sequences of 300 aa on ONE rank; include/snekmer_hip.h, skm_bucket_partition)
        # The check is COLLECTIVE: a rank whose shard is too large sends -1 counts through the first size gather instead of
        # raising alone, so that every rank raises (the others would otherwise wait for it inside the all-to-all).
        too_large = cap > self.MAX_SHARD_RESIDUES
        owners = self.columns == "owners"
        if too_large:
            d_counts.upload(np.full(d_counts.size, -1, dtype=np.int64))
        else:
            p_codes = self._need("p_codes", cap, self.code_dtype)
            p_rc = self._need("p_rc", cap, np.uint64)
            p_index = self._need("p_index", cap, np.uint32) if owners else None
            ctx.call("skm_bucket_partition", self.code_bits, G, _i64(loc.n), _i64(cap), _p(loc.rowptr.ptr),
                     _p(loc.codes.ptr), _p(loc.counts.ptr), _i64(lo), _p(p_codes.ptr), _p(p_rc.ptr), _p(d_counts.ptr), _p(None),
                     _p(p_index.ptr if owners else None))
        cmat = self._gather_sizes(d_counts, G)  # collective 1 + round trip 1: [src, dst] entry counts
        if (cmat < 0).any():
            bad = sorted(int(r) for r in np.nonzero((cmat < 0).any(axis=1))[0])
            raise ValueError(f"rank(s) {bad}: a shard of 2^30 residues or more; ShardedPipeline holds at most 2^30 - 1 per rank: use more ranks")
        loc.nnz = int(cmat[me, :].sum())
        self.nnz_total = int(cmat.sum())
        nrecv = int(cmat[:, me].sum())
        # 2. one grouped all-to-all (codes + posting words): every owner receives its k-mers' entries from all
        # ranks, sources in rank order
        r_codes = self._need("r_codes", nrecv, self.code_dtype)
        r_rc = self._need("r_rc", nrecv, np.uint64)
        ex.alltoallv_multi([p_codes, p_rc], [r_codes, r_rc], [cb, 8], cmat[me, :], cmat[:, me])  # collective 2
        # 3. owner: sort its share by code, compact postings, column starts, and either its answers (the owner-local column of
        # every entry it received, in the order received) or a hash table code -> column
        o_start = self._need("o_start", nrecv, np.uint32)
        o_post = self._need("o_post", nrecv, np.uint64)
        d_out4 = self._need("d_out4", 4, np.int64)
        if owners:
            o_ret = self._need("o_ret", nrecv, np.uint32)
            o_tkeys = o_tvals = None
        else:
            tcap = int(ctx.lib.skm_bucket_table_capacity(nrecv))
            o_tkeys = self._need("o_tkeys", tcap, self.code_dtype)
            o_tvals = self._need("o_tvals", tcap, np.uint32)
            o_ret = None
        ctx.call("skm_bucket_postings", self.code_bits, e.key_bits(self.lut.nsym, self.k), _i64(nrecv), _p(r_codes.ptr),
                 _p(r_rc.ptr), _p(d_out4.ptr), _p(None), _p(o_start.ptr), _p(o_post.ptr), _p(o_tkeys.ptr if o_tkeys else None),
                 _p(o_tvals.ptr if o_tvals else None), _p(o_ret.ptr if owners else None))
        # collective 3 + round trip 2: [rank, (distinct, shared columns, postings, table slots)]
        meta = self._gather_sizes(d_out4, 4)
        ncols, npost, tsize = meta[:, 1].copy(), meta[:, 2].copy(), meta[:, 3].copy()
        tot_cols, tot_post, tot_slots = int(ncols.sum()), int(npost.sum()), int(tsize.sum()) if not owners else 0
        # 4. one grouped all-gather: every rank gets all postings, column starts and row norms (and the tables); with
        # answers, the reverse of collective 2 carries one uint32 per entry back to where the entry came from (queued right
        # behind the all-gather: RCCL runs them back to back on the context's stream)
        b = self.basis or _ColumnMajor()
        b.ncols, b.ncols_shared = int(meta[:, 0].sum()), tot_cols
        b.post = self._need("post", tot_post, np.uint64)
        b.colptr = self._need("colptr", tot_cols + 1, np.uint32)
        a_start = self._need("a_start", tot_cols, np.uint32)
        rows64 = np.asarray(self.rows, dtype=np.int64)
        if owners:
            back = self._need("back", loc.nnz, np.uint32)
            ex.alltoallv_multi([o_ret], [back], [4], cmat[:, me], cmat[me, :])  # collective 4a: the answers
            ex.allgatherv_multi([o_post, o_start, rn], [b.post, a_start, self.rnorm], [8, 4, 4], [npost, ncols, rows64])  # 4b
        else:
            a_tkeys = self._need("a_tkeys", tot_slots, self.code_dtype)
            a_tvals = self._need("a_tvals", tot_slots, np.uint32)
            ex.allgatherv_multi([o_post, o_start, o_tkeys, o_tvals, rn], [b.post, a_start, a_tkeys, a_tvals, self.rnorm],
                                [8, 4, cb, 4, 4], [npost, ncols, tsize, tsize, rows64])  # collective 4
        ctx.call("skm_concat_colptr", G, ncols.ctypes.data_as(_p), npost.ctypes.data_as(_p), _p(a_start.ptr),
                 _p(b.colptr.ptr))
        self.basis = b
        # sizes of this step, for reporting (bench.py's per-stage rooflines)
        self.sizes = {"local_entries": int(loc.nnz), "owned_entries": nrecv, "owned_columns": int(ncols[me]),
                      "owned_postings": int(npost[me]), "owned_table_slots": 0 if owners else int(tsize[me]), "columns": tot_cols,
                      "postings": tot_post, "table_slots": tot_slots,
                      "alltoall_bytes_out": int((cmat[me, :].sum() - cmat[me, me]) * (cb + 8)),
                      "alltoall_bytes_in": int((cmat[:, me].sum() - cmat[me, me]) * (cb + 8)),
                      "answers_bytes_out": int((cmat[:, me].sum() - cmat[me, me]) * 4) if owners else 0,
                      "answers_bytes_in": int((cmat[me, :].sum() - cmat[me, me]) * 4) if owners else 0,
                      "allgather_bytes_in": int((tot_post - npost[me]) * 8 + (tot_cols - ncols[me]) * 4
                                                + (0 if owners else (tot_slots - tsize[me]) * (cb + 4)) + (self.n_total - nloc) * 4),
                      "column_ids": self.columns}
        # 5. columns of the local rows; the shard as rows [lo, hi) of an N-row matrix
        colidx = self._need("colidx", loc.nnz, np.uint32)
        if owners:
            groups = np.ascontiguousarray(cmat[me, :], dtype=np.int64)
            ctx.call("skm_colidx_from_owners", G, _i64(loc.nnz), _p(back.ptr), _p(p_index.ptr), groups.ctypes.data_as(_p),
                     ncols.ctypes.data_as(_p), _p(colidx.ptr))
        else:
            ctx.call("skm_colidx_lookup", self.code_bits, G, _i64(loc.nnz), _p(loc.codes.ptr), tsize.ctypes.data_as(_p),
                     ncols.ctypes.data_as(_p), _p(a_tkeys.ptr), _p(a_tvals.ptr), _p(colidx.ptr))
        ctx.call("skm_embed_rowptr", _i64(self.n_total), _i64(lo), _i64(nloc), _p(loc.rowptr.ptr), _p(self.rowptr_g.ptr))
        x = e.CountsCSR(ctx, self.n_total, loc.nnz, self.code_bits, self.rowptr_g, loc.codes, loc.counts, None)
        x.colidx = colidx
        self.x = x
        return x

    def _exchange_replicated(self, shard_batch):
        e, ctx = self.engine, self.ctx
        self.local = e.count_csr(ctx, shard_batch, self.lut, self.k, out=self.local)
        meta = self.ex.allgather_i64([self.local.nnz, e.csr_max_count(ctx, self.local)])
        nnz = meta[:, 0]
        narrow = int(meta[:, 1].max()) <= 255
        plan = plan_allgather(nnz, self.rows, np.dtype(self.code_dtype).itemsize, 1 if narrow else 4)
        self.ex.allgatherv(self.local.codes, plan["codes"], self.full.codes)
        if narrow:
            ctx.call("skm_narrow_u32_u8", _i64(self.local.nnz), _p(self.local.counts.ptr), _p(self.bytes_local.ptr))
            self.ex.allgatherv(self.bytes_local, plan["counts"], self.bytes_full)
            ctx.call("skm_widen_u8_u32", _i64(int(nnz.sum())), _p(self.bytes_full.ptr), _p(self.full.counts.ptr))
        else:
            self.ex.allgatherv(self.local.counts, plan["counts"], self.full.counts)
        self.ex.allgatherv(self.local.rowptr, plan["rowptr"], self.g_rowptr_local)
        h_rows = np.asarray(self.rows, dtype=np.int64)
        h_nnz = np.asarray(nnz, dtype=np.int64)
        ctx.call("skm_csr_concat_rowptr", len(self.rows), h_rows.ctypes.data_as(_p), h_nnz.ctypes.data_as(_p),
                 _p(self.g_rowptr_local.ptr), _p(self.full.rowptr.ptr))
        self.full.nnz = self.nnz_total = int(h_nnz.sum())
        self.basis = e.build_basis(ctx, self.full, self.lut.nsym, self.k, out=self.basis, elide_singletons=True)
        self.rnorm = e.row_norms(ctx, self.n_total, self.full.rowptr, self.full.counts, out=self.rnorm)
        self.x = self.full
        return self.full

    def step(self, shard_batch):
        """Rows [lo_r, hi_r) of the N x N cosine matrix as a dense float32 block in HBM."""
        e, ctx = self.engine, self.ctx
        self.exchange(shard_batch)
        lo, hi = self.bounds[self.rank]
        ld = (self.n_total + 3) // 4 * 4
        if self.out is None:
            self.out = ctx.empty((max(hi - lo, 1), ld), np.float32)
        b = self.basis
        e.cosine_matrix(ctx, self.x, self.rnorm, self.n_total, getattr(b, "ncols_shared", b.ncols), b.colptr, b.post,
                        self.rnorm, row0=lo, row1=hi, out=self.out, ld=ld)
        return self.out

    def step_topk(self, shard_batch, k: int, exclude_self: bool = True, cap_entries=None):
        """Reduced output for batches whose dense matrix cannot be stored (BASELINE configs[3],
        1 M sequences: SURVEY.md H6): the k best cosine neighbours of rows [lo_r, hi_r) against all
        N sequences, from the exact neighbour lists.  Returns (idx uint32[rows,k], score
        float32[rows,k], NeighborLists); idx is 0xFFFFFFFF where a row has fewer than k neighbours."""
        e, ctx = self.engine, self.ctx
        self.exchange(shard_batch)
        lo, hi = self.bounds[self.rank]
        b = self.basis
        if cap_entries is None:
            cap_entries = 16 * max(self.nnz_total // max(self.world, 1), 1) + (1 << 20)
        nb = e.gram_neighbors(ctx, self.x, self.rnorm, self.n_total, getattr(b, "ncols_shared", b.ncols), b.colptr, b.post,
                              self.rnorm, row0=lo, row1=hi, cap_entries=cap_entries)
        # overflow is a COLLECTIVE decision: a rank that raised alone would leave its peers waiting in the next
        # collective, so every rank learns every rank's count and all of them raise together
        over = self.ex.allgather_i64([nb.overflow_rows])[:, 0]
        if int(over.sum()):
            raise OverflowError(
                f"{int(over.sum())} rows (per rank: {over.tolist()}) exceed the neighbour-list capacity; raise cap_entries")
        idx, val = e.neighbors_topk(ctx, nb, self.rnorm, self.rnorm, k, exclude_self=exclude_self)
        return idx, val, nb
