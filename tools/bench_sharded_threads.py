#!/usr/bin/env python3
"""tools only: the sharded step (dist.ShardedPipeline, distributed basis) with G ranks as G threads on ONE GPU
(tests/inproc_world.py: the exchange executes the C library's byte plans with device copies).  One GPU does all G
ranks' work, so the wall time of a world step is about the SUM of the ranks' device work: wall / G estimates a rank's
compute at that G (no xGMI, no RCCL latency), and wall against the single-GPU step says how much extra total work the
sharded algorithm does.  Not a scaling measurement.   usage: bench_sharded_threads.py [n] [steps]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np

    from inproc_world import run_world
    from snekmer_amd import alphabet, engine
    from snekmer_amd.dist import ShardedPipeline, shard_bounds
    from snekmer_amd.synth import BASE_SEED, synth_families

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    lut = alphabet.build_lut("red6")
    res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
    out = {"n": n, "steps": steps, "worlds": []}
    for world in (1, 2, 4, 8):
        bounds = shard_bounds(n, world)
        gate = threading.Barrier(world)
        walls = [0.0] * world

        def body(rank, ctx, ex):
            lo, hi = bounds[rank]
            shard = engine.SeqBatch(ctx, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
            sp = ShardedPipeline(ctx, lut, 12, ex, bounds, int(off[-1]))
            for _ in range(2):
                sp.step(shard)
            ctx.sync()
            gate.wait()
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                sp.step(shard)
            ctx.sync()
            gate.wait()
            walls[rank] = (time.perf_counter() - t0) / steps * 1e3
            prof = ctx.profile_dump()
            return {"stage_ms": {k: v[1] / steps for k, v in prof.items()}, "sizes": sp.sizes}

        results, tw = run_world(world, body)
        kern = [sum(r["stage_ms"].values()) for r in results]
        out["worlds"].append({
            "ranks": world, "wall_ms_per_world_step": max(walls), "wall_over_ranks_ms": max(walls) / world,
            "rank0_stage_ms_sum": kern[0], "rank0_stages": {k: round(v, 3) for k, v in results[0]["stage_ms"].items() if v > 0.02},
            "bytes_moved_per_step": tw.bytes_moved // (steps + 2),
        })
    print(json.dumps(out))


if __name__ == "__main__":
    main()
