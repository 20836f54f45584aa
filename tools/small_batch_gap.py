#!/usr/bin/env python3
"""tools only: for small batches, how much of a step is kernels and how much is launch gaps / the host wait inside
skm_vectorize_csr.  Prints, per N, the wall ms per step and the sum of the per-kernel HIP-event times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import BASE_SEED, synth_families

    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    ctx = _hip.Context(0)
    lut = alphabet.build_lut("red6")
    out = []
    for n in (1000, 3383, 10000, 30000):
        res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 1)
        batch = engine.SeqBatch(ctx, res, off)
        p = engine.Pipeline(ctx, lut, 12)
        for _ in range(5):
            p.step(batch)
        ctx.sync()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            p.step(batch)
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(50):
            p.step(batch)
        prof = ctx.profile_dump()
        ctx.profile_enable(False)
        ksum = sum(v[1] for v in prof.values()) / 50
        out.append({"n": n, "wall_ms_per_step": wall, "kernel_sum_ms": ksum, "launches_per_step": sum(v[0] for v in prof.values()) / 50,
                    "stages": {k: round(v[1] / 50, 4) for k, v in prof.items()}})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
