#!/usr/bin/env python3
"""tools only: bench.py's `config4_one_rank_share` leg by itself (1 M sequences vectorized, one rank's 125 k x 1 M neighbour
lists, top-10, stage rooflines): python3 tools/bench_config4.py"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from snekmer_amd import _hip, alphabet, engine  # noqa: E402

if "red6" not in alphabet.ALPHABETS:
    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
args = types.SimpleNamespace(n=100000, length=300, k=12, alphabet="red6")
print(json.dumps(bench.config4_one_rank_share(_hip.default_context(), engine, alphabet, args), indent=1))
