#!/usr/bin/env python3
"""tools only: a few steps of the vectorize + N x N cosine pipeline on synth_skewed (what bench.py's `skewed_workload`
extra times), for profiler runs: tools/pmc_heavy.sh puts this under rocprofv3 --pmc.  usage: bench_skewed.py [n] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import BASE_SEED, synth_skewed

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    ctx = _hip.Context(0)
    res, off, _ = synth_skewed(n, seed=BASE_SEED + 12)
    batch = engine.SeqBatch(ctx, res, off)
    pipe = engine.Pipeline(ctx, alphabet.build_lut("red6"), 12, post32=os.environ.get("SKM_TOOL_POST32") == "1")
    pipe.step(batch)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(batch)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    prof = ctx.profile_dump()
    print(json.dumps({"n": n, "post_bits": pipe.basis.post_bits, "ms_per_step": dt,
                      "stages": {k: round(v[1] / steps, 3) for k, v in prof.items() if v[1] / steps > 0.05}}))


if __name__ == "__main__":
    main()
