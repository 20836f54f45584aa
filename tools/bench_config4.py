#!/usr/bin/env python3
"""BASELINE configs[3]: 1 M x 300 aa (red6 k=12), reduced output (top-10 cosine neighbours per row
from the exact neighbour lists; the dense 1 M x 1 M matrix would be 4 TB).

  python tools/bench_config4.py [n] [block]
      one GPU, one rank's share: vectorize all n sequences, neighbour lists + top-10 for the
      first `block` rows (default n/8) against all n columns.
  python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/bench_config4.py [n]
      the sharded job: each rank vectorizes n/G sequences, one RCCL all-gather of the CSR shards,
      then top-10 for its own row block (dist.ShardedPipeline.step_topk)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
lut = alphabet.build_lut("red6")
res, off, fam = synth_families(n, 300, family=100, seed=BASE_SEED + 3)

if world > 1:
    import torch
    import torch.distributed as dist

    from snekmer_amd.dist import RcclExchange, ShardedPipeline, shard_bounds

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _hip.Context(int(os.environ.get("LOCAL_RANK", "0")))
    uid = [RcclExchange.new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ex = RcclExchange(ctx, world, rank, uid[0])
    ex.allgather_i64([rank])
    warm_s, warm_r = ctx.zeros(8 * world, np.uint8), ctx.zeros(8 * world, np.uint8)
    ex.alltoallv(warm_s, [8] * world, warm_r, [8] * world)  # point-to-point channels set up before timing
    bounds = shard_bounds(n, world)
    lo, hi = bounds[rank]
    shard = engine.SeqBatch(ctx, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
    sp = ShardedPipeline(ctx, lut, 12, ex, bounds, int(off[-1]))
    for rnd in range(2):
        ctx.sync()
        dist.barrier()
        t0 = time.perf_counter()
        idx, val, nb = sp.step_topk(shard, 10, cap_entries=(hi - lo) * 6000)
        ctx.sync()
        dist.barrier()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    same = float(np.mean(fam[idx[:, 0].astype(np.int64) % n] == fam[lo:hi]))
    if rank == 0:
        print(json.dumps({"config": f"{n} x 300aa red6 k=12 sharded x{world}: all-gather CSR, top-10 per row", "step_ms": float(t.item()) * 1e3,
                          "sequences_per_s": n / float(t.item()), "nnz": sp.nnz_total, "basis_columns": sp.basis.ncols,
                          "rank0_top1_same_family_frac": same}))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)

block = int(sys.argv[2]) if len(sys.argv) > 2 else n // 8
ctx = _hip.Context(0)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.Pipeline(ctx, lut, 12)
for rnd in range(2):
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    pipe.vectorize(batch)
    ctx.sync()
    t_vec = time.perf_counter() - t0
    b = pipe.basis
    t0 = time.perf_counter()
    nb = engine.gram_neighbors(ctx, pipe.csr, pipe.rnorm, n, b.ncols, b.colptr, b.post, pipe.rnorm, row0=0, row1=block,
                               cap_entries=block * 6000, post_bits=b.post_bits, postcnt=b.postcnt)
    ctx.sync()
    t_nb = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, val = engine.neighbors_topk(ctx, nb, pipe.rnorm, pipe.rnorm, 10, exclude_self=True)
    t_top = time.perf_counter() - t0
    prof = ctx.profile_dump()
# sanity: the best neighbour of a row is a member of its family
same = float(np.mean(fam[idx[:, 0].astype(np.int64) % n] == fam[:block]))
print(json.dumps({
    "config": f"{n} x 300aa red6 k=12 on one MI355X; neighbour lists + top-10 for rows [0,{block}) x {n} columns",
    "vectorize_ms": t_vec * 1e3, "neighbors_ms": t_nb * 1e3, "topk_ms_incl_download": t_top * 1e3,
    "nnz": pipe.csr.nnz, "basis_columns": b.ncols, "list_entries": nb.total, "overflow_rows": nb.overflow_rows,
    "entries_per_row": nb.total / block, "top1_same_family_frac": same,
    "sequences_per_s_vectorize": n / t_vec, "rows_per_s_neighbors": block / t_nb,
    "kernels_ms": {k: v[1] for k, v in prof.items()},
}))
