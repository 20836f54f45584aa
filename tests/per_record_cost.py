#!/usr/bin/env python3
"""Cost of the per-record API an UNCHANGED rule body calls (rules/kmerize.smk:95,118: one
KmerVec.reduce_vectorize per sequence, twice), next to the reference-equivalent Python loop (oracle/ref_path.py,
test infrastructure) on this host, and next to the batch call that carries the real workloads."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_path
from snekmer_amd import alphabet, vectorize
from snekmer_amd.synth import synth_families, to_records

recs = [s for _, s in to_records(*synth_families(2000, 300, family=100, seed=3)[:2])]
out = {}
for name, k in (("hydro", 14), ("standard", 12)):
    kv = vectorize.KmerVec(name, k)
    table = alphabet.FULL_ALPHABETS[name]
    kv.reduce_vectorize(recs[0])
    t0 = time.perf_counter()
    for s in recs[:500]:
        kv.reduce_vectorize(s)
    per_rec = (time.perf_counter() - t0) / 500
    t0 = time.perf_counter()
    for s in recs[:500]:
        vectorize.reduce(s, name)
    per_reduce = (time.perf_counter() - t0) / 500
    t0 = time.perf_counter()
    for s in recs[:500]:
        ref_path.reduce_vectorize(s, k, table)
    per_ref = (time.perf_counter() - t0) / 500
    t0 = time.perf_counter()
    kv.reduce_vectorize_batch(recs)
    per_batch = (time.perf_counter() - t0) / len(recs)
    out[f"{name}_k{k}"] = {"reduce_vectorize_us_per_call": per_rec * 1e6, "reduce_us_per_call": per_reduce * 1e6,
                          "reference_python_loop_us_per_call": per_ref * 1e6,
                          "reduce_vectorize_batch_us_per_record": per_batch * 1e6}
print(json.dumps(out, indent=1))
