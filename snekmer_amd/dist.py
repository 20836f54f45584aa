"""dist: sequence-sharded vectorize + row-block cosine across the GPUs of one node.

The reference has no distributed layer; its only parallelism is one Snakemake job per FASTA
file (snekmer/rules/kmerize.smk:57-65).  Here sequences shard across ranks (one process per
GPU).  A k-mer's integer code is a pure function of (alphabet, k, window), so shards agree on
column identity without exchanging a dictionary; the single exchange is an all-gather of the
per-shard CSR rows, after which rank r computes rows [lo_r, hi_r) of the
similarity matrix against all N columns.

`plan_*` functions are pure host logic (covered by world_size-2 gloo tests on CPU);
`RcclExchange` is the device implementation over skm_allgatherv.
"""
import ctypes as C
from typing import List, Sequence, Tuple

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
        self._sizes = ctx.empty(world, np.int64)
        self._mine = ctx.empty(1, np.int64)

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
        if self._mine.size != vals.size:
            self._mine = self.ctx.empty(vals.size, np.int64)
            self._sizes = self.ctx.empty(self.world * vals.size, np.int64)
        self._mine.upload(vals)
        sizes = np.full(self.world, 8 * vals.size, dtype=np.int64)
        self.ctx.call("skm_allgatherv", _p(self._mine.ptr), sizes.ctypes.data_as(_p), _p(self._sizes.ptr))
        return self._sizes.download(self.world * vals.size).reshape(self.world, vals.size)

    def allgatherv(self, d_send, nbytes_per_rank: Sequence[int], d_recv):
        sizes = np.asarray(nbytes_per_rank, dtype=np.int64)
        self.ctx.call("skm_allgatherv", _p(d_send.ptr), sizes.ctypes.data_as(_p), _p(d_recv.ptr))


class ShardedPipeline:
    """vectorize the local shard, all-gather CSR, then cosine for the local row block."""

    def __init__(self, ctx, lut, k: int, exchange, bounds: Sequence[Tuple[int, int]], total_residues: int):
        from . import engine

        self.engine = engine
        self.ctx, self.lut, self.k, self.ex = ctx, lut, k, exchange
        self.bounds = list(bounds)
        self.rank = exchange.rank
        self.n_total = self.bounds[-1][1]
        self.rows = [hi - lo for lo, hi in self.bounds]
        bits = lut.code_bits(k)
        self.code_dtype = np.uint32 if bits == 32 else np.uint64
        cap = total_residues + 1
        self.local = None
        self.bytes_local = ctx.empty(cap, np.uint8)   # counts of the local shard, one byte each
        self.bytes_full = ctx.empty(cap, np.uint8)
        self.g_rowptr_local = ctx.empty(self.n_total + len(self.bounds), np.int64)
        full = engine.CountsCSR(ctx, self.n_total, 0, bits, ctx.empty(self.n_total + 1, np.int64),
                                ctx.empty(cap, self.code_dtype), ctx.empty(cap, np.uint32), None)
        self.full = full
        self.basis = None
        self.rnorm = None
        self.out = None

    def exchange(self, shard_batch):
        """vectorize the local shard and all-gather the CSR shards: afterwards `self.full` (all N
        rows), `self.basis` (postings of the full matrix) and `self.rnorm` are set on every rank."""
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
        self.full.nnz = int(h_nnz.sum())
        self.basis = e.build_basis(ctx, self.full, self.lut.nsym, self.k, out=self.basis, elide_singletons=True)
        self.rnorm = e.row_norms(ctx, self.n_total, self.full.rowptr, self.full.counts, out=self.rnorm)
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
        e.cosine_matrix(ctx, self.full, self.rnorm, self.n_total, b.ncols, b.colptr, b.post, self.rnorm,
                        row0=lo, row1=hi, out=self.out, ld=ld)
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
        nb = e.gram_neighbors(ctx, self.full, self.n_total, b.ncols, b.colptr, b.post, row0=lo, row1=hi,
                              cap_entries=cap_entries)
        if nb.overflow_rows:
            raise OverflowError(f"{nb.overflow_rows} rows exceed the neighbour-list capacity; raise cap_entries")
        idx, val = e.neighbors_topk(ctx, nb, self.rnorm, self.rnorm, k, exclude_self=exclude_self)
        return idx, val, nb
