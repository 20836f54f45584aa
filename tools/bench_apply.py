#!/usr/bin/env python3
"""tools only: bench.py's `apply_chain` leg by itself (learn aggregation by column, CSR by family, fused apply epilogue,
materialised block for comparison) at the bench workload: python3 tools/bench_apply.py [n]"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from snekmer_amd import _hip, alphabet, engine  # noqa: E402
from snekmer_amd.synth import BASE_SEED  # noqa: E402

if "red6" not in alphabet.ALPHABETS:
    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
args = types.SimpleNamespace(n=int(sys.argv[1]) if len(sys.argv) > 1 else 100000, length=300, k=12, alphabet="red6")
print(json.dumps(bench.apply_chain(_hip.default_context(), engine, alphabet, args, BASE_SEED + 2), indent=1))
