#!/usr/bin/env python3
"""tools only: engine.Pipeline (one stream) against engine.OverlappedPipeline (next batch vectorized on a second
context beside this batch's Gram) on the bench workload: ms per step of each, interleaved, and a bit-for-bit check of
the result (row sums + non-zero counts of the whole matrix, and 64 rows)."""
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import BASE_SEED, synth_families

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    def groups(var):
        v = os.environ.get(var)
        return tuple(int(x) for x in v.split("-")) if v else None

    ctx = _hip.Context(0, cu_groups=groups("SKM_AB_MAIN_GROUPS"))  # (a confined main context confines the baseline too)
    lut = alphabet.build_lut("red6")
    res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
    batch = engine.SeqBatch(ctx, res, off)
    # SKM_AB_EXTRA_CTX=n: n idle contexts (streams) alive beside the two that work: HIP maps streams onto a few hardware
    # queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams on one queue do not overlap
    idle = [_hip.Context(0) for _ in range(int(os.environ.get("SKM_AB_EXTRA_CTX", "0")))]
    a = engine.Pipeline(ctx, lut, 12)
    # SKM_AB_SIDE_GROUPS=a-b: the side context confined to those CU groups (default: OverlappedPipeline's own choice, 0-3);
    # SKM_AB_SIDE_GROUPS=all: an unconfined side context
    side = None
    if os.environ.get("SKM_AB_SIDE_GROUPS") == "all":
        side = _hip.Context(0)
    elif os.environ.get("SKM_AB_SIDE_GROUPS"):
        side = _hip.Context(0, cu_groups=groups("SKM_AB_SIDE_GROUPS"))
    # SKM_AB_FRACTION=f: the neighbour lists of the last f * n rows are built on the side context
    frac = float(os.environ["SKM_AB_FRACTION"]) if os.environ.get("SKM_AB_FRACTION") else None
    # SKM_AB_DEPTH=d: d batches in flight on side contexts
    depth = int(os.environ.get("SKM_AB_DEPTH", "1"))
    b = engine.OverlappedPipeline(ctx, lut, 12, side_ctx=side, side_list_fraction=frac, depth=depth)
    b.out = a.step(batch)  # share the 40 GB result buffer
    ctx.sync()
    ld = a.out.shape[1]
    sums_a, nnz_a = engine.matrix_row_stats(ctx, a.out, n, n, ld)
    rows = np.linspace(0, n - 1, 64).astype(np.int64)
    want = [a.out.download(n, offset=int(r) * ld) for r in rows]
    ctx.call("skm_memset", __import__("ctypes").c_void_p(a.out.ptr), 0, __import__("ctypes").c_size_t(a.out.nbytes))
    for _ in range(depth):
        b.prefetch(batch)
    b.step(batch)
    b.sync()
    sums_b, nnz_b = engine.matrix_row_stats(ctx, b.out, n, n, ld)
    same = bool((sums_a == sums_b).all() and (nnz_a == nnz_b).all() and
                all((b.out.download(n, offset=int(r) * ld) == w).all() for r, w in zip(rows, want)))
    out = {"n": n, "steps": steps, "depth": depth, "fraction": b.fraction, "identical": same, "runs": []}
    for rep in range(3):
        a.step(batch)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            a.step(batch)
        ctx.sync()
        ta = (time.perf_counter() - t0) / steps * 1e3
        b.step(batch)
        b.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.step(batch)
        b.sync()
        tb = (time.perf_counter() - t0) / steps * 1e3
        out["runs"].append({"pipeline_ms": ta, "overlapped_ms": tb})
    # drain the prefetched batch so that the process ends with nothing queued
    for _ in range(depth):
        b.step(None)
    b.sync()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
