#!/usr/bin/env python3
"""bench.py — vectorize + all-pairs cosine throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch that is already resident in
HBM: recode + k-mer count (CSR) -> observed basis / postings -> row norms -> N x N float32
cosine, all outputs left in HBM.  Workload at any --gpus: BASELINE.json configs[2]
(100k x 300 aa, alphabet=red6, k=12); with N > 1 the same 100k sequences are sharded by rows
(strong scaling): each rank vectorizes its shard, the ranks build the postings of the full
matrix together (RCCL all-to-all by k-mer owner, all-gather of the postings; snekmer_amd/dist.py),
then each rank computes its row block of the matrix.

Prints ONE JSON line on rank 0 (contract: see repo prompt / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(alphabet_name, k, seed, n_sample):
    """The reference-equivalent numpy path (oracle/ref_path.py: np.isin basis pass, O(N*B) count
    projection, float64 normalise+dot) timed on one host core over a bounded sample of the same
    synthetic workload.  The oracle is only the reported baseline here, never the measured path."""
    from oracle import c_oracle, ref_path
    from snekmer_amd import alphabet
    from snekmer_amd.synth import synth_families, to_records

    res, off, _ = synth_families(n_sample, 300, family=100, seed=seed)
    recs = to_records(res, off)
    table = alphabet.FULL_ALPHABETS[alphabet_name]
    t0 = time.perf_counter()
    ref_path.vectorize_and_cosine(recs, k, table)
    dt = time.perf_counter() - t0
    # sparse C restatement on a larger sample, for scale (still one core)
    n_c = 4000
    res, off, _ = synth_families(n_c, 300, family=100, seed=seed)
    lut = alphabet.build_lut(alphabet_name)
    t1 = time.perf_counter()
    rowptr, codes, counts, first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
    b, _, _, _, col = c_oracle.basis(rowptr, codes, counts, first)
    c_oracle.cosine_rows(rowptr, col, counts, len(b), np.arange(n_c))
    dt_c = time.perf_counter() - t1
    return {
        "value": n_sample / dt,
        "unit": "sequences/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_sample} x 300aa synthetic families, red6 k={k}: oracle/ref_path.py "
        f"(reference-equivalent numpy path, cost grows ~N^2) took {dt:.1f}s",
        "host_cores_available": os.cpu_count(),
        "sparse_c_oracle": {"value": n_c / dt_c, "unit": "sequences/s", "cores": 1,
                            "sample": f"{n_c} x 300aa, oracle/kmer_oracle.c took {dt_c:.2f}s"},
    }


def reserve_stdout() -> int:
    """gloo and RCCL print banners on stdout (RCCL through C stdio, flushed as late as process
    exit); stdout is reserved for the one JSON line.  Point fd 1 at stderr for the whole run and
    return a private descriptor of the real stdout for that line."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def load_pmc_traffic():
    """Per-launch HBM bytes of the dominant kernel from the committed PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("k_cosine_write_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--k", type=int, default=12)
    ap.add_argument("--alphabet", default="red6")
    ap.add_argument("--cpu-sample", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    real_stdout = reserve_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import BASE_SEED, synth_families

    if "red6" not in alphabet.ALPHABETS:
        alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)

    # SKM_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (gloo control plane, RCCL exchange,
    # ShardedPipeline) even with one rank, so that a 1-GPU box can exercise it.
    sharded = world > 1 or os.environ.get("SKM_BENCH_FORCE_SHARDED") == "1"
    dist = None
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed (gloo) is control plane only: id broadcast, barriers, max-over-ranks.
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)

    ctx = _hip.Context(local_rank)
    lut = alphabet.build_lut(args.alphabet)
    seed = BASE_SEED + 2
    res, off, _ = synth_families(args.n, args.length, family=100, seed=seed)
    n_total = args.n
    residues_total = int(off[-1])

    if not sharded:
        batch = engine.SeqBatch(ctx, res, off)
        pipe = engine.Pipeline(ctx, lut, args.k)
        step = lambda: pipe.step(batch)
        rows_local = n_total
    else:
        from snekmer_amd.dist import RcclExchange, ShardedPipeline, shard_bounds

        uid = [RcclExchange.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ex = RcclExchange(ctx, world, rank, uid[0])
        # first collectives: RCCL sets up its ring and its point-to-point channels here, outside the timed region
        ex.allgather_i64([rank])
        warm_s, warm_r = ctx.zeros(8 * world, np.uint8), ctx.zeros(8 * world, np.uint8)
        ex.alltoallv(warm_s, [8] * world, warm_r, [8] * world)
        ctx.sync()
        bounds = shard_bounds(n_total, world)
        lo, hi = bounds[rank]
        shard = engine.SeqBatch(ctx, res[off[lo] : off[hi]], off[lo : hi + 1] - off[lo])
        pipe = ShardedPipeline(ctx, lut, args.k, ex, bounds, residues_total)
        step = lambda: pipe.step(shard)
        rows_local = hi - lo

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_dump()
    ctx.profile_enable(False)

    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        launches, strip_ms = prof.get("k_cosine_write", (0, 0.0))
        strip_avg_ms = strip_ms / max(launches, 1)
        nnz = pipe.nnz_total if sharded else pipe.csr.nnz
        ld = (n_total + 3) // 4 * 4
        # algorithmic bytes of one k_cosine_write launch (DESIGN.md "Kernels"): the float32 output
        # rows it must write; the sparse neighbour lists it reads are <1% of that and not counted.
        # (One launch per step; written so that it stays right if a step ever splits the launch.)
        algo_bytes = rows_local * ld * 4 * args.steps / max(launches, 1)
        achieved = algo_bytes / (strip_avg_ms * 1e-3) / 1e9 if strip_avg_ms > 0 else 0.0
        line = {
            "metric": "sequences/sec vectorize+pairwise-cosine, 100k x 300aa k=12",
            "value": n_total / (elapsed / args.steps),
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: {n_total} x {args.length}aa synthetic protein families "
                f"(100/family, 10% substitutions, shuffled), alphabet={args.alphabet} k={args.k}, "
                "recode+count(CSR)+basis+full NxN float32 cosine resident in HBM",
                "n_sequences": n_total,
                "residues": residues_total,
                "nnz": nnz,
                "basis_columns": pipe.basis.ncols,
                "parallelism": f"row-sharded x{world}, postings built by k-mer owner (RCCL all-to-all + all-gathers)"
                if sharded else "single GPU",
            },
            "residues_per_s": residues_total / (elapsed / args.steps),
            "stage_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
            "roofline": {
                "kernel": "k_cosine_write",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_pmc_traffic(),
                "launches": launches,
                "avg_launch_ms": strip_avg_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not sharded and args.alphabet == "red6":
            # red6 is a benchmark alphabet; the nearest reference alphabet (`standard`,
            # 7^12 needs uint64 codes) is timed next to it on the same sequences, outside the timed region
            lut7 = alphabet.build_lut("standard")
            pipe7 = engine.Pipeline(ctx, lut7, args.k)
            pipe7.out = pipe.out  # share the 40 GB result buffer
            pipe7.step(batch)
            ctx.sync()
            t1 = time.perf_counter()
            for _ in range(3):
                pipe7.step(batch)
            ctx.sync()
            dt7 = (time.perf_counter() - t1) / 3
            line["reference_alphabet_check"] = {
                "alphabet": "standard", "k": args.k, "code_bits": 64, "ms_per_step": dt7 * 1e3,
                "sequences_per_s": n_total / dt7, "nnz": pipe7.csr.nnz, "basis_columns": pipe7.basis.ncols,
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.alphabet, args.k, seed, args.cpu_sample)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
