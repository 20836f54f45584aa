#!/usr/bin/env python3
"""tools only: an external yardstick for the dense i8 MFMA cosine kernel (k_cosine_dense_i8_v5 / _v4).

Times the vendor library's int8 x int8 -> int32 GEMM (torch._int_mm -> hipBLASLt on ROCm) at the benchmark's shape
(N = M = 32768, K = 16384: hydro k=14, full basis) next to the product kernel through the C-ABI, same operands, same
process.  The library never enters the product; this answers "is 0.56 of peak the kernel or the hardware?".

    python3 tools/gemm_yardstick.py [N] [K]    -> one JSON line
    python3 tools/gemm_yardstick.py counts      -> the same with the operand the product really multiplies: the int8 k-mer
                                                   count matrix of the synthetic families (hydro k=14, N = 32768), which is
                                                   mostly zeros (the chip clocks higher on it than on random values)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    counts = len(sys.argv) > 1 and sys.argv[1] == "counts"
    n = int(sys.argv[1]) if len(sys.argv) > 1 and not counts else 32768
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    out = {"shape": f"N = M = {n}, K = {k}, int8 x int8 -> int32", "peak_POPS": 5.0,
           "operand": "k-mer counts of synthetic families (hydro k=14), mostly zeros" if counts else "uniform random values 0..2"}
    ops = 2.0 * n * n * k
    count_host = None
    if counts:
        try:
            import torch

            torch.cuda.init()  # before this process creates its own HIP context
        except Exception:  # noqa: BLE001
            pass
        from snekmer_amd import _hip, alphabet, engine
        from snekmer_amd.synth import BASE_SEED, synth_families

        ctx0 = _hip.Context(0)
        res_, off_, _ = synth_families(n, 300, family=100, seed=BASE_SEED + 6)
        pipe = engine.DensePipeline(ctx0, alphabet.build_lut("hydro"), 14)
        pipe.step(engine.SeqBatch(ctx0, res_, off_))
        assert pipe.kdim == k
        count_host = pipe.dense.download().reshape(-1)[: n * k].reshape(n, k).copy()
        del pipe

    # ---- vendor library
    try:
        import torch

        dev = torch.device("cuda:0")
        g = torch.Generator(device="cpu").manual_seed(1)
        a = torch.from_numpy(count_host).to(dev) if counts else torch.randint(0, 3, (n, k), dtype=torch.int8, generator=g).to(dev)
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
        host = count_host if counts else (np.random.default_rng(1).integers(0, 3, size=(n, k))).astype(np.int8)

    # ---- the product kernel, same operand
    from snekmer_amd import _hip, engine

    ctx = _hip.Context(0)
    d = ctx.to_device(host)
    rn = engine.row_norms_i8(ctx, n, k, d)
    res = engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn)
    ctx.sync()
    ref_sum = None
    for variant, label in (("7", "rectangular launch (every tile), v4: both operands through LDS"),
                           ("11", "rectangular launch, v5: A through LDS, B straight into registers (what a rectangular X, Y call runs)"),
                           ("6", "symmetric launch (X is Y: upper tiles + mirrored stores), v4"),
                           (None, "symmetric launch, v5 (what DensePipeline runs)")):
        _hip.set_option("SKM_DENSE_VARIANT", variant)
        engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn, out=res)
        ctx.sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            engine.cosine_dense_i8(ctx, n, n, k, d, d, rn, rn, out=res)
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # every variant is exact: a strided sample of the result must be identical across them
        sample = res.download()[:: max(n // 509, 1), :: max(n // 1021, 1)].astype(np.float64)
        digest = float(sample.sum()), float((sample * sample).sum())
        if ref_sum is None or variant == "6":  # the symmetric launches are compared with the first symmetric one
            ref_sum = digest
        out.setdefault("product", []).append({"call": f"skm_cosine_dense_i8: {label}", "ms": ms, "POPS_by_full_problem": ops / (ms * 1e-3) / 1e15,
                                              "equals_first_variant_of_its_kind_on_sample": digest == ref_sum,
                                              "note": "includes the float32 scaling epilogue and the operand re-tiling pass"})
    _hip.set_option("SKM_DENSE_VARIANT", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
