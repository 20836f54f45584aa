#!/usr/bin/env python3
"""tools only: an external yardstick for the dense i8 MFMA cosine kernel (k_cosine_dense_i8_v4).

Times the vendor library's int8 x int8 -> int32 GEMM (torch._int_mm -> hipBLASLt on ROCm) at the benchmark's shape
(N = M = 32768, K = 16384: hydro k=14, full basis) next to the product kernel through the C-ABI, same operands, same
process.  The library never enters the product; this answers "is 0.56 of peak the kernel or the hardware?".

    python3 tools/gemm_yardstick.py [N] [K]    -> one JSON line
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    out = {"shape": f"N = M = {n}, K = {k}, int8 x int8 -> int32", "peak_POPS": 5.0}
    ops = 2.0 * n * n * k

    # ---- vendor library
    try:
        import torch

        dev = torch.device("cuda:0")
        g = torch.Generator(device="cpu").manual_seed(1)
        a = torch.randint(0, 3, (n, k), dtype=torch.int8, generator=g).to(dev)
        bt = a.t()  # [k, n] view of the same operand: C = A @ A^T, the X-is-Y problem without the symmetry shortcut
        for name, rhs in (("A @ A^T (B column-major view)", bt), ("A @ B (B row-major copy)", bt.contiguous())):
            try:
                c = torch._int_mm(a, rhs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    c = torch._int_mm(a, rhs)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                out.setdefault("library", []).append({"call": f"torch._int_mm: {name}", "ms": ms, "POPS": ops / (ms * 1e-3) / 1e15,
                                                      "frac_of_peak": ops / (ms * 1e-3) / 1e15 / 5.0})
                del c
            except Exception as exc:  # noqa: BLE001
                out.setdefault("library", []).append({"call": f"torch._int_mm: {name}", "error": str(exc)[:300]})
        host = a.cpu().numpy()
        del a, bt
        torch.cuda.empty_cache()
    except Exception as exc:  # noqa: BLE001
        out["library_error"] = str(exc)[:300]
        host = (np.random.default_rng(1).integers(0, 3, size=(n, k))).astype(np.int8)

    # ---- the product kernel, same operand
    from snekmer_amd import _hip, engine

    ctx = _hip.Context(0)
    d = ctx.to_device(host)
    rn = engine.row_norms_i8(ctx, n, k, d)
    res = engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn)
    ctx.sync()
    for variant, label in (("7", "rectangular launch (every tile)"), (None, "symmetric launch (X is Y: upper tiles + mirrored stores)")):
        if variant:
            os.environ["SKM_DENSE_VARIANT"] = variant
        else:
            os.environ.pop("SKM_DENSE_VARIANT", None)
        engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn, out=res)
        ctx.sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn, out=res)
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out.setdefault("product", []).append({"call": f"skm_cosine_dense_i8: {label}", "ms": ms, "POPS_by_full_problem": ops / (ms * 1e-3) / 1e15,
                                              "note": "includes the float32 scaling epilogue and the operand re-tiling pass"})
    os.environ.pop("SKM_DENSE_VARIANT", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
