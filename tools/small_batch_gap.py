#!/usr/bin/env python3
"""tools only: for small batches, how much of a step is kernels and how much is launch gaps / host waits.  Prints, per
N, the wall ms per step, the sum of the per-stage HIP-event times, the launches per step and the HBM floor of the step
(4 bytes per result cell at the 8 TB/s spec).  SKM_COSINE_PATH / SNEKMER_HIP_LIB select kernels / builds as usual."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(ctx, lut, k, res, off, reps=200, pipeline=None):
    from snekmer_amd import engine

    batch = engine.SeqBatch(ctx, res, off)
    p = (pipeline or engine.Pipeline)(ctx, lut, k)
    walls = {}
    for mode in (False, "auto"):  # eager launches, then whole-step HIP-graph replay (the default)
        p.graphs = mode
        for _ in range(5):
            p.step(batch)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.step(batch)
        ctx.sync()
        walls[mode] = (time.perf_counter() - t0) / reps * 1e3
    wall = walls[False]
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(50):
        p.step(batch)
    prof = ctx.profile_dump()
    ctx.profile_enable(False)
    n = batch.n
    ld = (n + 3) // 4 * 4
    floor_ms = 4.0 * n * ld / 8e12 * 1e3
    return {"n": n, "wall_ms_per_step": wall, "wall_ms_per_step_graph_replay": walls["auto"], "graph_replays": p.graph_replays, "kernel_sum_ms": sum(v[1] for v in prof.values()) / 50,
            "launches_per_step": sum(v[0] for v in prof.values()) / 50, "hbm_floor_ms": floor_ms, "frac": floor_ms / wall,
            "stages": {kk: round(v[1] / 50, 4) for kk, v in prof.items()}}


def main():
    from snekmer_amd import _hip, alphabet
    from snekmer_amd.synth import BASE_SEED, synth_families

    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    ctx = _hip.Context(0)
    lut = alphabet.build_lut("red6")
    out = []
    for n in [int(x) for x in (sys.argv[1:] or (1000, 3383, 10000, 30000))]:
        res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 1)
        out.append(measure(ctx, lut, 12, res, off))
    # the reference's CI proteome at its CI configuration (solvacc k=8: 6561 columns): dense route (int8 GEMM)
    from snekmer_amd.io import read_fasta_packed

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data", "UP000322080_2603819.fasta")
    if os.path.exists(path) and not sys.argv[1:]:
        _, res, off = read_fasta_packed(path)
        for route in ("auto", False):
            from snekmer_amd import engine

            r = measure(ctx, alphabet.build_lut("solvacc"), 8, res, off, pipeline=lambda c, l, k: engine.Pipeline(c, l, k, dense_route=route))
            r["workload"] = f"UP000322080 solvacc k=8, dense_route={route}"
            out.append(r)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
