#!/usr/bin/env python3
"""Diagnostic: time k_cosine_strip variants in one process (interleaved rounds).
SKM_COSINE_ABLATE: 0 real kernel, 1 no accumulate, 2 no global stores, 3 plain (not nt) stores."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine
from snekmer_amd.synth import BASE_SEED, synth_families

alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _hip.Context(0)
lut = alphabet.build_lut("red6")
res, off, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 2)
batch = engine.SeqBatch(ctx, res, off)
pipe = engine.Pipeline(ctx, lut, 12)
pipe.step(batch)
ctx.sync()
ctx.profile_enable(True)
gres = {}
for rnd in range(3):
    for name, env in (("full", {}), ("w8x1024", {"SKM_WRITE_VARIANT": "1"}), ("w16x1024", {"SKM_WRITE_VARIANT": "2"}),
                      ("w1x4096", {"SKM_WRITE_VARIANT": "3"}), ("w2x2048", {"SKM_WRITE_VARIANT": "4"}),
                      ("w2x4096", {"SKM_WRITE_VARIANT": "5"}), ("w1x2048", {"SKM_WRITE_VARIANT": "6"}), ("no_pairs", {"SKM_GRAM_ABLATE": "1"}), ("no_emit", {"SKM_GRAM_ABLATE": "2"}), ("no_insert", {"SKM_GRAM_ABLATE": "4"}), ("no_load", {"SKM_GRAM_ABLATE": "5"}),
                      ("v1", {"SKM_GRAM_VARIANT": "1"}), ("v2", {"SKM_GRAM_VARIANT": "2"}), ("v3", {"SKM_GRAM_VARIANT": "3"}),
                      ("v4", {"SKM_GRAM_VARIANT": "4"}), ("v5", {"SKM_GRAM_VARIANT": "5"}), ("v6", {"SKM_GRAM_VARIANT": "6"})):
        for k in ("SKM_GRAM_ABLATE", "SKM_GRAM_VARIANT", "SKM_WRITE_VARIANT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx.profile_reset()
        pipe.cosine()
        gres.setdefault(name, []).append((ctx.profile_read("k_gram_sparse")[1], ctx.profile_read("k_cosine_write")[1], ctx.profile_read("k_cosine_strip")[1], ctx.profile_read("k_gram_sparse_big")[1]))
for k in ("SKM_GRAM_ABLATE", "SKM_GRAM_VARIANT", "SKM_WRITE_VARIANT"):
    os.environ.pop(k, None)
for name, v in gres.items():
    print(f"gram[{name}]: " + ", ".join(f"gram {a:.2f} big {d:.2f} write {b:.2f} cursor {c:.2f}" for a, b, c, d in v))
results = {}
for rnd in range(0):
    for abl in (0, 1, 2, 3):
        os.environ["SKM_COSINE_ABLATE"] = str(abl)
        ctx.profile_reset()
        pipe.cosine()
        cnt, ms = ctx.profile_read("k_cosine_strip")
        results.setdefault(abl, []).append(ms)
os.environ["SKM_COSINE_ABLATE"] = "0"
for abl, v in results.items():
    print(f"ABL={abl}: min {min(v):.3f} ms  median {sorted(v)[len(v)//2]:.3f} ms  all {['%.2f' % x for x in v]}")

# phase stamps of the Gram kernel (diagnostic build, exact results)
import ctypes as C
os.environ["SKM_GRAM_ABLATE"] = "3"
pipe.cosine()
ticks = (C.c_ulonglong * 8)()
ctx.lib.skm_debug_gram_phases.argtypes = [C.c_void_p, C.c_void_p]
ctx.lib.skm_debug_gram_phases(ctx.handle, ticks)
os.environ.pop("SKM_GRAM_ABLATE")
names = ["zero+rowptr", "tasks+scan", "pair loop", "hist+scan", "emit"]
tot = sum(ticks[:5]) or 1
print("gram phases (ticks per row, share): " + ", ".join(f"{n} {ticks[i]/pipe.csr.n:.0f} ({100*ticks[i]/tot:.0f}%)" for i, n in enumerate(names)))
