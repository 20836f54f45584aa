#!/usr/bin/env python3
"""tools only: one round of tests/fuzz_parity.py by seed, with a Python stack dump if it does not finish (a hang shows
where the host waits).  usage: fuzz_one.py <seed> [seconds before the dump]"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity  # noqa: E402
from snekmer_amd import _hip  # noqa: E402
from snekmer_amd import alphabet as A  # noqa: E402

seed = int(sys.argv[1])
faulthandler.dump_traceback_later(float(sys.argv[2]) if len(sys.argv) > 2 else 60.0, exit=True)
if "red6" not in A.ALPHABETS:
    A.register_alphabet("red6", A.RED6_GROUPS)
ctx = _hip.default_context()
fuzz_parity.one_round(ctx, seed, verbose=True)
print("round ok", flush=True)
