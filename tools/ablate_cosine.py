#!/usr/bin/env python3
"""Diagnostic timings of the cosine stage on one MI355X (interleaved rounds in one process).

  SKM_GRAM_ABLATE   1 no pair loop   2 no emit   4 loads but no hash insert   5 second half of every list only   6/7/8 without the bin of lists of 17+ / 5-16 / 1-4 postings
                    3 exact kernel with shader-clock stamps per phase
  SKM_COSINE_ABLATE (cursor kernel) 1 no accumulate   2 no global stores   3 plain (not nt) stores
  SKM_COSINE_PATH=cursor  the general fallback kernel for every strip

Results of ablated builds are invalid by construction; only their times are of interest.  The
ablations exist only in the diagnostic library (`make -C snekmer_amd/csrc diag` ->
libsnekmer_hip_diag.so, built with -DSKM_DIAG), which this script loads instead of the product."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip, alphabet, engine

_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsnekmer_hip_diag.so")
if not os.path.exists(_hip.LIB_PATH):
    sys.exit("build the diagnostic library first: make -C snekmer_amd/csrc diag")
from snekmer_amd.synth import BASE_SEED, synth_families

KNOBS = ("SKM_GRAM_ABLATE", "SKM_COSINE_ABLATE", "SKM_COSINE_PATH", "SKM_GRAM_SHAPE")
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
cases = [
    ("full", {}),
    ("gram: no pair loop", {"SKM_GRAM_ABLATE": "1"}),
    ("gram: no emit", {"SKM_GRAM_ABLATE": "2"}),
    ("gram: no hash insert", {"SKM_GRAM_ABLATE": "4"}),
    ("gram: half of every list", {"SKM_GRAM_ABLATE": "5"}),
    ("gram: no lists of 17+", {"SKM_GRAM_ABLATE": "6"}),
    ("gram: no lists of 5-16", {"SKM_GRAM_ABLATE": "7"}),
    ("gram: no lists of 1-4", {"SKM_GRAM_ABLATE": "8"}),
    ("gram: no column-start gathers, no pair loop", {"SKM_GRAM_ABLATE": "9"}),
    ("gram shape 16x2 (exact)", {"SKM_GRAM_SHAPE": "1"}),
    ("gram shape 16x1 (exact)", {"SKM_GRAM_SHAPE": "2"}),
    ("gram shape 32x1 (exact)", {"SKM_GRAM_SHAPE": "3"}),
    ("gram shape 64x1 (exact)", {"SKM_GRAM_SHAPE": "4"}),
    ("cursor kernel everywhere", {"SKM_COSINE_PATH": "cursor"}),
    ("cursor: no accumulate", {"SKM_COSINE_ABLATE": "1"}),
    ("cursor: no stores", {"SKM_COSINE_ABLATE": "2"}),
]
rows = {}
for rnd in range(3):
    for name, env in cases:
        for k in KNOBS:
            _hip.set_option(k, env.get(k))
        ctx.profile_reset()
        pipe.cosine()
        rows.setdefault(name, []).append(tuple(ctx.profile_read(k)[1] for k in
                                               ("k_gram_sparse", "k_gram_sparse_big", "k_cosine_write", "k_cosine_strip")))
for k in KNOBS:
    _hip.set_option(k, None)
print(f"{'case':28s} gram   big    write  cursor   (ms, median of 3)")
for name, v in rows.items():
    med = [sorted(x[i] for x in v)[1] for i in range(4)]
    print(f"{name:28s} " + " ".join(f"{x:6.2f}" for x in med))

_hip.set_option("SKM_GRAM_ABLATE", 3)
pipe.cosine()
ticks = (C.c_ulonglong * 8)()
ctx.lib.skm_debug_gram_phases.argtypes = [C.c_void_p, C.c_void_p]
ctx.lib.skm_debug_gram_phases(ctx.handle, ticks)
_hip.set_option("SKM_GRAM_ABLATE", None)
names = ["zero+rowptr", "tasks+scan", "pair loop", "emit"]
tot = sum(ticks[:4]) or 1
print("gram phases (ticks per row, share of workgroup lifetime): " +
      ", ".join(f"{nm} {ticks[i] / pipe.csr.n:.0f} ({100 * ticks[i] / tot:.0f}%)" for i, nm in enumerate(names)))
