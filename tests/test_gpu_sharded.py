"""GPU parity of the SHARDED path (dist.ShardedPipeline + skm_shard.hip + the C exchange plans) with G > 1 ranks on
one GPU: every rank is a thread of this process (tests/inproc_world.py), the exchange executes the C library's own
byte plans with device copies.  The reference has no distributed layer (one Snakemake job per FASTA file,
snekmer/rules/kmerize.smk:57-65); the bar is bit-identity with the single-GPU pipeline, which the parity suite pins
against the oracle at the same sizes (test_gpu_parity.py: config 3 and config 4).
"""
import sys

import numpy as np
import pytest

from helpers import ensure_red6
from inproc_world import progress, run_world

pytestmark = pytest.mark.gpu

ensure_red6()


@pytest.fixture(scope="module")
def ctx():
    from snekmer_amd import _hip

    return _hip.default_context()


def _canonical_lists(start, length, jj, dot):
    """Neighbour lists -> one sorted uint64 word per entry, (row << 47 | j << 27 | dot): equal arrays <=> the same
    neighbour set with the same exact dot for every row.  A single-key sort (no argsort): the 1 M-sequence blocks hold
    more than 10^8 entries."""
    length = length.astype(np.int64)
    assert (length != 0xFFFFFFFF).all() and len(length) < (1 << 17)
    tot = int(length.sum())
    first = np.cumsum(length) - length
    src = np.repeat(start.astype(np.int64) - first, length) + np.arange(tot, dtype=np.int64)
    key = np.repeat(np.arange(len(length), dtype=np.uint64) << np.uint64(47), length)
    j, d = jj[src], dot[src]
    assert tot == 0 or (int(j.max()) < (1 << 20) and 0 < int(d.min()) and int(d.max()) < (1 << 27))
    key |= j.astype(np.uint64) << np.uint64(27)
    key |= d.astype(np.uint64)
    key.sort()
    return (key,)


def _assert_topk_equal(idx_a, val_a, idx_b, val_b):
    """Same scores everywhere; same neighbours wherever the score is not tied with the next-best candidates."""
    assert idx_a.shape == idx_b.shape and (val_a == val_b).all()
    diff = idx_a != idx_b
    if diff.any():
        k = val_a.shape[1]
        tied = np.zeros_like(diff)
        tied[:, 1:] |= val_a[:, 1:] == val_a[:, :-1]
        tied[:, :-1] |= val_a[:, :-1] == val_a[:, 1:]
        tied[:, k - 1] = True  # the last place may tie with the first candidate left out
        assert (tied | ~diff).all()
        assert diff.mean() < 0.02


def _run_sharded(world, lut, k, res, off, bounds, body_extra):
    from snekmer_amd import engine
    from snekmer_amd.dist import ShardedPipeline

    def body(rank, rctx, ex):
        lo, hi = bounds[rank]
        shard = engine.SeqBatch(rctx, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
        sp = ShardedPipeline(rctx, lut, k, ex, bounds, int(off[-1]))
        return body_extra(rank, rctx, sp, shard)

    return run_world(world, body)


@pytest.mark.parametrize("n", [18000, 26000])
def test_sharded_two_thread_ranks_owner_sort_tile_shapes(ctx, n):
    """Shards of 2.7 and 3.9 M residues: the stable grouping by owner (the one-sweep sort on one-byte keys) takes its 1024 x 12 /
    1024 x 16 tiles there (skm_onesweep.h: the tile grows with the capacity so that a pass's grid is resident at once).  Every
    rank's row block against the single-GPU result: totals, norms, per-row sums and non-zero counts of the whole block,
    32 rows bit for bit."""
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.dist import shard_bounds
    from snekmer_amd.synth import synth_families

    lut, k, world = A.build_lut("red6"), 12, 2
    res, off, _ = synth_families(n, 300, family=40, seed=91)
    bounds = shard_bounds(n, world)

    def extra(rank, rctx, sp, shard):
        lo, hi = bounds[rank]
        out = sp.step(shard)
        ld = out.shape[1]
        sums, nnz = engine.matrix_row_stats(rctx, out, hi - lo, n, ld)
        rows = np.linspace(0, hi - lo - 1, 32).astype(np.int64)
        return {"sums": sums, "nnzrow": nnz, "rows": rows, "sample": np.stack([out.download(n, offset=int(r) * ld) for r in rows]),
                "nnz": sp.nnz_total, "ncols": sp.basis.ncols, "rnorm": sp.rnorm.download(n)}

    results, _ = _run_sharded(world, lut, k, res, off, bounds, extra)
    ref = engine.Pipeline(ctx, lut, k)
    S = ref.step(engine.SeqBatch(ctx, res, off))
    ld = S.shape[1]
    sums, nnzrow = engine.matrix_row_stats(ctx, S, n, n, ld)
    rn = ref.rnorm.download(n)
    for rank, r in enumerate(results):
        lo, hi = bounds[rank]
        assert (r["nnz"], r["ncols"]) == (ref.csr.nnz, ref.basis.ncols)
        assert (r["rnorm"] == rn).all()
        assert (r["nnzrow"] == nnzrow[lo:hi]).all() and (r["sums"] == sums[lo:hi]).all()
        for q, row in enumerate(r["rows"]):
            assert (r["sample"][q] == S.download(n, offset=int(lo + row) * ld)).all(), (rank, int(row))


@pytest.mark.parametrize("world,n,name,by_residues", [(8, 2400, "red6", False), (4, 1800, "standard", True), (8, 5, "red6", False),
                                                      (3, 1000, "hydro", False)])
def test_sharded_threads_small_dense_block_and_topk(ctx, world, n, name, by_residues):
    """Up to 8 ranks (threads) through ShardedPipeline.step and step_topk: stacked row blocks, norms, entry and column
    totals bit-identical to the single-GPU pipeline; uint32 and uint64 codes; a world larger than the batch (empty
    shards); hydro k=12 (4096 possible columns: every owner's table is dense, long posting lists)."""
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.dist import shard_bounds, shard_bounds_by_residues
    from snekmer_amd.synth import synth_families

    lut, k, topk = A.build_lut(name), 12, 5
    res, off, _ = synth_families(n, 300, family=30, seed=77)
    bounds = shard_bounds_by_residues(off, world) if by_residues else shard_bounds(n, world)

    def extra(rank, rctx, sp, shard):
        lo, hi = bounds[rank]
        for _ in range(2):  # the second step reuses every buffer
            out = sp.step(shard)
        block = out.download().reshape(out.shape)[: hi - lo, :n].copy()
        idx, val, nb = sp.step_topk(shard, topk)
        return {"block": block, "idx": idx, "val": val, "nnz": sp.nnz_total, "ncols": sp.basis.ncols,
                "rnorm": sp.rnorm.download(n), "lists": nb.host()}

    results, tw = _run_sharded(world, lut, k, res, off, bounds, extra)
    ref = engine.Pipeline(ctx, lut, k)
    batch = engine.SeqBatch(ctx, res, off)
    S = ref.step(batch)
    S = S.download().reshape(S.shape)[:n, :n]
    rn = ref.rnorm.download(n)
    b = ref.basis
    for rank, r in enumerate(results):
        lo, hi = bounds[rank]
        assert (r["nnz"], r["ncols"]) == (ref.csr.nnz, b.ncols)
        assert (r["rnorm"] == rn).all()
        assert (r["block"] == S[lo:hi]).all()
        if hi == lo:
            continue
        nb = engine.gram_neighbors(ctx, ref.csr, ref.rnorm, n, b.ncols, b.colptr, b.post, ref.rnorm, row0=lo, row1=hi, post_bits=b.post_bits, postcnt=b.postcnt)
        want, got = _canonical_lists(*nb.host()), _canonical_lists(*r["lists"])
        for w, g in zip(want, got):
            assert (w == g).all()
        idx, val = engine.neighbors_topk(ctx, nb, ref.rnorm, ref.rnorm, topk, exclude_self=True)
        _assert_topk_equal(idx, val, r["idx"], r["val"])
    if world > 1 and n > world:
        assert tw.bytes_moved > 0


def test_config4_sharded_code_path_8_ranks_1m_sequences(ctx):
    """BASELINE configs[3] (1 M x 300 aa, red6 k=12, sharded 8 ways) through the SHARDED code path at full size: 8 ranks
    (threads, one GPU) each count their 125 k sequences, group 36 M entries by owner, exchange them by the C library's
    all-to-all plan, sort and emit the postings + hash table of their 1/8 of the k-mers, all-gather the five arrays by
    the all-gather plan, look up the columns of their rows and produce exact neighbour lists + top-10 of their row
    block against all 1 M sequences.  Against the single-GPU pipeline on the same batch (itself pinned to the oracle
    at this size by test_config4_one_rank_share_125k_rows_vs_1m): norms of all rows, entry / column totals, and for
    the blocks of ranks 0 and 5 every neighbour set, every exact integer dot and the top-10."""
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.dist import shard_bounds
    from snekmer_amd.synth import synth_families

    _, _, mem = ctx.device_info()
    if mem < 200 * 2**30:
        pytest.skip("needs ~120 GB of HBM (8 ranks' buffers on one device)")
    lut, k, n, world, topk = A.build_lut("red6"), 12, 1_000_000, 8, 10
    check = (0, 5)
    import faulthandler

    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)  # a stuck run says where (pytest shows stderr on failure)
    try:
        _config4_body(ctx, lut, k, n, world, topk, check)
    finally:
        faulthandler.cancel_dump_traceback_later()


def _config4_body(ctx, lut, k, n, world, topk, check):
    from snekmer_amd import engine
    from snekmer_amd.dist import shard_bounds
    from snekmer_amd.synth import synth_families

    progress("config 4: generating 1 M sequences")
    res, off, _ = synth_families(n, 300, family=100, seed=20250523 + 3)
    bounds = shard_bounds(n, world)

    def extra(rank, rctx, sp, shard):
        progress("shard resident; step_topk")
        idx, val, nb = sp.step_topk(shard, topk, cap_entries=(bounds[rank][1] - bounds[rank][0]) * 4000)
        progress(f"step_topk done: {sp.sizes}")
        out = {"nnz": sp.nnz_total, "ncols": sp.basis.ncols, "shared": sp.basis.ncols_shared, "local_nnz": sp.local.nnz,
               "overflow": nb.overflow_rows, "entries": nb.total}
        if rank in check:
            out.update(idx=idx, val=val, lists=nb.host(), rnorm=sp.rnorm.download(n))
        return out

    results, tw = _run_sharded(world, lut, k, res, off, bounds, extra)
    progress("8 ranks done; single-GPU reference")
    ref = engine.Pipeline(ctx, lut, k)
    ref.vectorize(engine.SeqBatch(ctx, res, off))
    b = ref.basis
    rn = ref.rnorm.download(n)
    assert sum(r["local_nnz"] for r in results) == ref.csr.nnz
    for rank, r in enumerate(results):
        lo, hi = bounds[rank]
        assert (r["nnz"], r["ncols"], r["overflow"]) == (ref.csr.nnz, b.ncols, 0)
        if rank not in check:
            continue
        assert (r["rnorm"] == rn).all()
        nb = engine.gram_neighbors(ctx, ref.csr, ref.rnorm, n, b.ncols, b.colptr, b.post, ref.rnorm, row0=lo, row1=hi, cap_entries=(hi - lo) * 4000,
                                   post_bits=b.post_bits, postcnt=b.postcnt)
        assert nb.overflow_rows == 0 and nb.total == r["entries"]
        want, got = _canonical_lists(*nb.host()), _canonical_lists(*r["lists"])
        for w, g in zip(want, got):
            assert (w == g).all()
        idx, val = engine.neighbors_topk(ctx, nb, ref.rnorm, ref.rnorm, topk, exclude_self=True)
        _assert_topk_equal(idx, val, r["idx"], r["val"])
        del nb
        progress(f"rank {rank}'s block equal")
    # the exchange moved what the design says it moves: every entry once (12 B, 7/8 of them off-rank) in the
    # all-to-all, then 7 copies of every owner's arrays in the all-gather
    assert tw.bytes_moved > ref.csr.nnz * 12 * 7 // 8


def test_oversized_shard_is_refused_by_every_rank(ctx):
    """A shard beyond the per-rank limit must not leave the other ranks waiting inside the next collective: the rank that
    finds it sends -1 counts through the first size gather and EVERY rank raises (here the limit is lowered on rank 1 only)."""
    from snekmer_amd import alphabet as A
    from snekmer_amd.dist import shard_bounds
    from snekmer_amd.synth import synth_families

    lut, k, world, n = A.build_lut("red6"), 12, 3, 600
    res, off, _ = synth_families(n, 300, family=20, seed=5)
    bounds = shard_bounds(n, world)

    def extra(rank, rctx, sp, shard):
        if rank == 1:
            sp.MAX_SHARD_RESIDUES = 1000
        try:
            sp.step(shard)
        except ValueError as exc:
            return str(exc)
        return None

    results, _ = _run_sharded(world, lut, k, res, off, bounds, extra)
    assert all(r is not None and "rank(s) [1]" in r for r in results), results
