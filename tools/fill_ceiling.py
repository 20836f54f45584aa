#!/usr/bin/env python3
"""Pure-write ceiling: time hipMemsetAsync over a 40 GB buffer (wall clock around a stream sync)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snekmer_amd import _hip

ctx = _hip.Context(0)
nbytes = 40_000_000_000
buf = ctx.empty(nbytes, np.uint8)
for rnd in range(4):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.call("skm_memset", C.c_void_p(buf.ptr), 0, C.c_size_t(nbytes))
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"memset 40 GB: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s")
