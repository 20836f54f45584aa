#!/usr/bin/env python3
"""bench.py — vectorize + all-pairs cosine throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch that is already resident in
HBM: recode + k-mer count (CSR) -> observed basis / postings -> row norms -> N x N float32
cosine, all outputs left in HBM.  On one GPU the steps are those of a STREAM of batches
(engine.OverlappedPipeline, --pipeline overlapped, the default): while a step's cosine runs on the main
stream, the next batch's recode / count / sort / basis run on a second stream confined to half of the
compute units; every timed step holds one complete vectorize and one complete cosine, nothing is cached
and the results are bit-identical to the one-stream pipeline's, whose step (--pipeline single) is carried
in the same line as `single_batch_step`.  Workload at any --gpus: BASELINE.json configs[2]
(100k x 300 aa, alphabet=red6, k=12); with N > 1 the same 100k sequences are sharded by rows
(strong scaling): each rank vectorizes its shard, the ranks build the postings of the full
matrix together (RCCL all-to-all by k-mer owner, all-gather of the postings; snekmer_amd/dist.py),
then each rank computes its row block of the matrix.

Prints ONE JSON line on rank 0 (contract: see repo prompt / DESIGN.md section "Measurement").
Besides the contract keys the line carries, all measured in this same run after the timed region:
  single_batch_step      one batch on one stream (engine.Pipeline): ms/step, stage times, the writer alone
  stage_rooflines        every stage of that single-batch step with its algorithmic bytes and fraction of HBM peak
  config2                BASELINE configs[1] (10k sequences) ms/step
  host_to_result_ms      H2D of the packed batch + one step (the PCIe-inclusive figure; never `value`)
  reference_alphabet_check   the nearest reference alphabet (standard k=12, uint64 codes)
  config5_count_dense    BASELINE configs[4]: 100k x 2^20 uint16 dense count scatter (210 GB)
  dense_mfma             the i8 MFMA cosine GEMM at hydro k=14, N = 32768
  apply_chain            SURVEY 8(f) f1/f2: learn aggregation + fused apply epilogue, 100 k queries x 1000 family totals
  config4_one_rank_share BASELINE configs[3]: 1 M sequences vectorized + one rank's 125 k x 1 M neighbour lists + top-10
  overlapped_equals_single   the timed object's last result against the one-stream pipeline's, whole matrix, in this run
  cpu_baseline           reference-equivalent numpy path at several N with a quadratic fit, and the
                         sparse C restatement on one core and on all cores of this GPU's host share
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

# BASELINE.json's metric, verbatim
BASELINE_METRIC = "sequences/sec vectorize+pairwise-cosine, 100k\u00d7300aa k=12; 1/2/4/8 GPU"

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_PEAK_TOPS = 5000.0  # same guide: i8 MFMA = 2x the bf16 rate per clock -> ~5 POPS dense


# --------------------------------------------------------------------------------- CPU baselines
def cpu_point_child(n, k, alphabet_name, seed):
    """Child-process mode (--cpu-point N): time oracle/ref_path.py, the reference-equivalent numpy path
    (np.isin basis pass, O(N*B) count projection, float64 normalise+dot), on N synthetic sequences with
    one thread.  Never touches the GPU."""
    from oracle import ref_path
    from snekmer_amd import alphabet
    from snekmer_amd.synth import synth_families, to_records

    if "red6" not in alphabet.ALPHABETS:
        alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    res, off, _ = synth_families(n, 300, family=100, seed=seed)
    recs = to_records(res, off)
    table = alphabet.FULL_ALPHABETS[alphabet_name]
    t0 = time.perf_counter()
    out, counts, S = ref_path.vectorize_and_cosine(recs, k, table)
    dt = time.perf_counter() - t0
    print(json.dumps({"n": n, "seconds": dt, "basis": int(counts.shape[1]), "checksum": float(S.sum())}))


def cpu_baseline(alphabet_name, k, seed, points, budget_s, n_sparse):
    """(1) oracle/ref_path.py at every N of `points`, one single-threaded process per point (run side by
    side to bound the wall time), with a least-squares fit t = a*N^2 + b*N; (2) the sparse C restatement
    (oracle/kmer_oracle.c: CSR counts, sort-unique basis, exact sparse Gram + float32 scaling, every row of
    the N x N result produced) on one core and, with OpenMP, on all cores of this GPU's share of the host.
    The oracle is only the reported baseline here, never the measured path."""
    from oracle import c_oracle
    from snekmer_amd import alphabet
    from snekmer_amd.synth import synth_families

    env = dict(os.environ)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        env[var] = "1"
    env["HIP_VISIBLE_DEVICES"] = ""  # the children have no business on the GPU
    t_start = time.perf_counter()
    procs = [
        (n, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-point", str(n), "--k", str(k),
                              "--alphabet", alphabet_name], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
        for n in points
    ]

    # sparse C restatement while the children run (it uses other cores)
    lut = alphabet.build_lut(alphabet_name)
    threads = c_oracle.host_threads(ignore_omp_env=not points)  # multi-rank runs: the launcher pins OMP_NUM_THREADS=1

    def sparse(n, nthreads):
        res, off, _ = synth_families(n, 300, family=100, seed=seed)
        t0 = time.perf_counter()
        rowptr, codes, counts, first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off, threads=nthreads)
        t1 = time.perf_counter()
        b, _, _, _, col = c_oracle.basis(rowptr, codes, counts, first, threads=nthreads)
        t2 = time.perf_counter()
        total = c_oracle.cosine_all(rowptr, col, counts, len(b), threads=nthreads)
        t3 = time.perf_counter()
        return {"value": n / (t3 - t0), "unit": "sequences/s", "cores": nthreads,
                "sample": f"{n} x 300aa, oracle/kmer_oracle.c: count {t1 - t0:.2f}s + basis {t2 - t1:.2f}s + "
                          f"N x N cosine rows {t3 - t2:.2f}s (each float32 row produced in a per-thread buffer, not kept)",
                "checksum": total}

    sparse_all = sparse(n_sparse, threads)
    sparse_one = sparse(min(n_sparse, 20000), 1)

    measured = []
    for n, p in procs:
        left = budget_s - (time.perf_counter() - t_start)
        try:
            out, _ = p.communicate(timeout=max(left, 1.0))
            rec = json.loads(out.decode().strip().splitlines()[-1])
            measured.append({"n": n, "seconds": rec["seconds"], "sequences_per_s": n / rec["seconds"], "basis": rec["basis"]})
        except Exception:
            p.kill()
            p.wait()
            measured.append({"n": n, "seconds": None, "note": f"not finished within the {budget_s:.0f}s budget"})
    good = [m for m in measured if m.get("seconds")]
    fit = None
    if len(good) >= 2:
        N = np.asarray([m["n"] for m in good], dtype=np.float64)
        T = np.asarray([m["seconds"] for m in good], dtype=np.float64)
        A = np.stack([N * N, N], axis=1)
        (a, b), *_ = np.linalg.lstsq(A, T, rcond=None)
        fit = {"model": "seconds = a*N^2 + b*N (least squares through the measured points)", "a": float(a), "b": float(b),
               "EXTRAPOLATED_seconds_at_10k": float(a * 1e8 + b * 1e4), "EXTRAPOLATED_seconds_at_100k": float(a * 1e10 + b * 1e5),
               "EXTRAPOLATED_sequences_per_s_at_100k": float(1e5 / (a * 1e10 + b * 1e5)) if a * 1e10 + b * 1e5 > 0 else None,
               "note": "extrapolation only: the dense float64 N x |basis| arrays of the reference "
                       "(rules/kmerize.smk:112) need ~170 GB at N = 10k, so the path cannot run at the benchmarked size"}
    if not points:
        return {"value": sparse_one["value"], "unit": "sequences/s", "cores": 1, "kind": "port", "sample": sparse_one["sample"],
                "note": "multi-rank run: only the sparse C restatement is timed here; the reference-equivalent numpy points and "
                        "their fit are a rank-0 leg of the 1-GPU run (python3 bench.py)",
                "sparse_all_cores": sparse_all, "sparse_one_core": sparse_one, "host_cores_visible": os.cpu_count(),
                "host_cores_used_for_all_cores": threads}
    head = good[-1] if good else {"n": 0, "sequences_per_s": 0.0, "seconds": 0.0}
    return {
        "value": head["sequences_per_s"],
        "unit": "sequences/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{head['n']} x 300aa synthetic families, {alphabet_name} k={k}: oracle/ref_path.py "
                  f"(reference-equivalent numpy path, one thread) took {head['seconds']:.1f}s; cost grows ~N^2, see points/fit",
        "points": measured,
        "fit": fit,
        "sparse_all_cores": sparse_all,
        "sparse_one_core": sparse_one,
        "host_cores_visible": os.cpu_count(),
        "host_cores_used_for_all_cores": threads,
    }


_T0 = time.perf_counter()


def note(msg: str) -> None:
    """Progress on stderr (stdout carries the one JSON line)."""
    sys.stderr.write(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}\n")
    sys.stderr.flush()


def reserve_stdout() -> int:
    """gloo and RCCL print banners on stdout (RCCL through C stdio, flushed as late as process
    exit); stdout is reserved for the one JSON line.  Point fd 1 at stderr for the whole run and
    return a private descriptor of the real stdout for that line."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def kernel_source_sha() -> str:
    """Hash of the sources of the dominant kernel: profiles/pmc_traffic.json records the hash it was
    measured on, so a stale PMC figure is never attached to a changed kernel."""
    h = hashlib.sha256()
    for f in ("skm_cosine_csr.hip", "skm_gram_kernel.h"):
        with open(os.path.join(ROOT, "snekmer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc_traffic():
    """Per-launch HBM bytes of the dominant kernel from the committed PMC summary (rocprofv3 --pmc
    WRITE_SIZE / FETCH_SIZE in separate passes, tools/profile_round.sh); None when that summary was
    taken on different kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except Exception:
        return None, "no PMC summary committed"
    if rec.get("source_sha16") != kernel_source_sha():
        return None, "PMC summary in profiles/ predates the current kernel sources; re-run tools/profile_round.sh"
    return rec.get("k_cosine_write_bytes_per_launch"), rec.get("note", "rocprofv3 --pmc, separate passes")


def pmc_child(args, seed):
    """Body of one rocprofv3 --pmc pass (see live_pmc_traffic): the bench workload, one warm-up and two steps."""
    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import synth_families

    if "red6" not in alphabet.ALPHABETS:
        alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    ctx = _hip.Context(0)
    res, off, _ = synth_families(args.n, args.length, family=100, seed=seed)
    batch = engine.SeqBatch(ctx, res, off)
    lut = alphabet.build_lut(args.alphabet)
    if args.pipeline == "overlapped":
        # the timed object: its k_cosine_write launches (two per step: the rows whose lists the main context built, then the
        # rows whose lists a side context built) are the launches roofline.algorithmic_bytes_per_launch is stated for
        op = engine.OverlappedPipeline(ctx, lut, args.k)
        op.prefetch(batch)
        for _ in range(3):
            op.step(batch)
        op.step(None)
        op.sync()
        return
    pipe = engine.Pipeline(ctx, lut, args.k)
    for _ in range(3):
        pipe.step(batch)
    ctx.sync()


def live_pmc_traffic(args, budget_s=150.0):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two child processes of this script under
    `rocprofv3 --pmc WRITE_SIZE` and `rocprofv3 --pmc FETCH_SIZE` (separate passes, as MI355X_MICROARCH.md
    prescribes; values are KiB per dispatch, FETCH_SIZE doubled for gfx950's wide streaming reads).  Called before
    this process touches the GPU; rocprofv3 launches this interpreter's own binary (sys.executable) directly.  Returns (bytes or None, note)."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    t0 = time.perf_counter()
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            left = budget_s - (time.perf_counter() - t0)
            if left < 20:
                return None, "live PMC passes ran out of their time budget"
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--n", str(args.n), "--length", str(args.length),
                   "--k", str(args.k), "--alphabet", args.alphabet, "--pipeline", args.pipeline]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed: {r.stderr.decode(errors='replace')[-200:]}"
            vals = []
            for root, _, files in os.walk(os.path.join(tmp, counter)):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        with open(os.path.join(root, f)) as fh:
                            for row in csv.DictReader(fh):
                                if row["Counter_Name"] == counter and re.search(r"\bk_cosine_write\b", row["Kernel_Name"]):
                                    vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"rocprofv3 --pmc {counter}: no k_cosine_write dispatch in the output"
            got[counter] = (sum(vals) / len(vals) * 1024.0, len(vals))
    w, f = got["WRITE_SIZE"], got["FETCH_SIZE"]
    return w[0] + 2.0 * f[0], (f"measured in this run on the timed pipeline (--pipeline {args.pipeline}): rocprofv3 --pmc WRITE_SIZE "
                               f"({w[0] / 1e9:.2f} GB/launch, {w[1]} launches) and --pmc FETCH_SIZE (2 x {f[0] / 1e9:.3f} GB), averaged over "
                               f"the k_cosine_write dispatches, separate child passes, {time.perf_counter() - t0:.0f} s")


def self_launch(n_ranks: int) -> int:
    """Run this script as `n_ranks` ranks under torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a
    free port) and relay rank 0's JSON line.  The child is a fresh interpreter started with subprocess (no exec of a
    process that has used the GPU: this parent never opens the device).  Returns the launcher's exit code."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note("self-launch: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for raw in proc.stdout:
        text = raw.decode(errors="replace").rstrip("\n")
        if text.startswith("{") and '"metric"' in text:
            line = text
        elif text:
            sys.stderr.write(text + "\n")
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    if rc != 0:
        note(f"self-launch: the {n_ranks}-rank job exited with code {rc}")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--k", type=int, default=12)
    ap.add_argument("--alphabet", default="red6")
    ap.add_argument("--pipeline", choices=("overlapped", "single"), default="overlapped",
                    help="one GPU: 'overlapped' = engine.OverlappedPipeline, the steps of a stream of batches (the next batch's count / sort / "
                         "basis run on a second, CU-confined stream beside this batch's cosine; every step holds one complete vectorize and "
                         "one complete cosine); 'single' = engine.Pipeline, one batch at a time on one stream")
    ap.add_argument("--cpu-points", default="250,500,1000,2000",
                    help="N values at which the reference-equivalent numpy path is timed (BASELINE.md section 3)")
    ap.add_argument("--cpu-budget-s", type=float, default=210.0, help="wall-time bound of the CPU-baseline leg")
    ap.add_argument("--cpu-sparse-n", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip config 2 / config 5 / MFMA / host-to-result extras")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from profiles/pmc_traffic.json instead of two "
                    "rocprofv3 --pmc child passes in this run")
    ap.add_argument("--cpu-point", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from snekmer_amd.synth import BASE_SEED

    seed = BASE_SEED + 2
    if args.cpu_point:
        cpu_point_child(args.cpu_point, args.k, args.alphabet, seed)
        return
    if args.pmc_child:
        pmc_child(args, seed)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: start the ranks ourselves, BEFORE anything in this process
        # touches the GPU (the parent never does: it relays the one JSON line and the exit code)
        sys.exit(self_launch(args.gpus))

    real_stdout = reserve_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        note(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used")
        args.gpus = world

    from snekmer_amd import _hip, alphabet, engine
    from snekmer_amd.synth import synth_families

    if "red6" not in alphabet.ALPHABETS:
        alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)

    # roofline.traffic: PMC passes as child processes, before this process opens the device
    live_traffic = None
    under_profiler = any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if (world == 1 and rank == 0 and not args.no_live_pmc and not under_profiler
            and os.environ.get("SKM_BENCH_FORCE_SHARDED") != "1"):
        note("rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE child passes")
        live_traffic = live_pmc_traffic(args)
        note(f"PMC traffic: {live_traffic[1]}")

    # SKM_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (gloo control plane, RCCL exchange,
    # ShardedPipeline) even with one rank, so that a 1-GPU box can exercise it.
    sharded = world > 1 or os.environ.get("SKM_BENCH_FORCE_SHARDED") == "1"
    dist = None
    if sharded:
        # a rank that waits for ever (a collective whose peer died, a transport that never connects) must not hold its GPU
        # until somebody else's time limit: after 15 minutes the rank prints where it stands and exits
        import faulthandler

        faulthandler.dump_traceback_later(900, exit=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed (gloo) is control plane only: id broadcast, barriers, max-over-ranks.
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)

    ndev = _hip.device_count()
    if sharded and 0 < ndev <= local_rank:
        # fewer GPUs than ranks (a 1-GPU box asked for --gpus 2): share devices, so that the failure is RCCL's own
        # refusal of two ranks on one device and not an argument error
        note(f"rank {rank}: {ndev} GPU(s) visible for {world} ranks; using device {local_rank % ndev}")
        local_rank %= ndev
    ctx = _hip.Context(local_rank)
    lut = alphabet.build_lut(args.alphabet)
    res, off, _ = synth_families(args.n, args.length, family=100, seed=seed)
    n_total = args.n
    residues_total = int(off[-1])

    op = None
    if not sharded:
        batch = engine.SeqBatch(ctx, res, off)
        pipe = engine.Pipeline(ctx, lut, args.k)
        step = lambda: pipe.step(batch)
        rows_local = n_total
        if args.pipeline == "overlapped":
            # a stream of batches THAT ARRIVES FROM THE HOST: three different batches (three seeds of the same generator) cycle
            # through pinned staging buffers into three recycled pairs of device buffers, one asynchronous upload per step on a
            # copy context of its own (engine.BatchUploader), ordered against the side context's vectorize by events on the
            # device: no allocation, no fill and no host wait per batch.  Every job of the reference starts from a file
            # (snekmer/rules/kmerize.smk:89-129).  The one-stream pipeline runs once first: the 40 GB result buffer is
            # shared, and the extras measure it by itself afterwards
            pipe.step(batch)
            ctx.sync()
            stream_batches = [(res, off)] + [synth_families(args.n, args.length, family=100, seed=seed + 101 * j)[:2] for j in (1, 2)]
            uploader = engine.BatchUploader(ctx, max(int(r.size) for r, _ in stream_batches), args.n, slots=3)
            op = engine.OverlappedPipeline(ctx, lut, args.k)
            op.out = pipe.out
            stream_pos = [0]

            def next_upload():
                r, o = stream_batches[stream_pos[0] % len(stream_batches)]
                stream_pos[0] += 1
                return uploader.upload(r, o)

            # uploads run ONE BATCH AHEAD of the vectorize that reads them: the copy of batch i + 2 is queued at the start
            # of step i (it runs beside step i's kernels), the side context vectorizes batch i + 1 - uploaded during step
            # i - 1 - without waiting for a copy.  (With the upload queued in the step that vectorizes it the side stream
            # starts 1.2 ms later, and it is nearly as full as the main one: 9.82 ms per step instead of 9.4.)
            op.prefetch(next_upload())
            ahead = [next_upload()]

            def step():
                nxt = ahead[0]
                ahead[0] = next_upload()
                return op.step(nxt)
    else:
        from snekmer_amd.dist import RcclExchange, ShardedPipeline, shard_bounds

        uid = [RcclExchange.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ex = RcclExchange(ctx, world, rank, uid[0])
        # first collectives: RCCL sets up its ring and its point-to-point channels here, outside the timed region
        ex.allgather_i64([rank])
        warm_s, warm_r = ctx.zeros(8 * world, np.uint8), ctx.zeros(8 * world, np.uint8)
        ex.alltoallv(warm_s, [8] * world, warm_r, [8] * world)
        ctx.sync()
        bounds = shard_bounds(n_total, world)
        lo, hi = bounds[rank]
        shard = engine.SeqBatch(ctx, res[off[lo] : off[hi]], off[lo : hi + 1] - off[lo])
        pipe = ShardedPipeline(ctx, lut, args.k, ex, bounds, residues_total)
        step = lambda: pipe.step(shard)
        rows_local = hi - lo

    def barrier():
        ctx.sync()
        if op is not None:
            op.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    note(f"input resident ({n_total} sequences, {residues_total} residues); warm-up")
    for _ in range(args.warmup):
        step()
    barrier()
    # the side contexts time the vectorize stages of the next batch, the Gram of its last rows and (their scratch holds those
    # lists) the writer launches for those rows
    prof_ctxs = [ctx] + (op.contexts() if op is not None else [])
    for c in prof_ctxs:
        c.profile_enable(True)
        c.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {}
    for c in prof_ctxs:
        for key, (calls, ms) in c.profile_dump().items():
            have = prof.get(key, (0, 0.0))
            prof[key] = (have[0] + calls, have[1] + ms)
        c.profile_enable(False)
    note(f"timed region: {elapsed / args.steps * 1e3:.3f} ms/step on rank {rank}")
    identity = None
    stream_info = None
    if op is not None:
        last = op.step(None)  # the batch still prefetched: nothing stays queued
        op.sync()
        # which batch that was: upload number stream_pos - 2 of the cycle (one more is uploaded ahead and never vectorized);
        # the one-stream reference below runs on a resident copy of the same residues
        last_res, last_off = stream_batches[(stream_pos[0] - 2) % len(stream_batches)]
        stream_info = {"distinct_batches": len(stream_batches), "uploads_in_timed_region": args.steps,
                       "host_waits_for_a_staging_buffer": uploader.host_waits,
                       "h2d_bytes_per_step": int(last_res.nbytes + 64 + last_off.nbytes)}
        batch = engine.SeqBatch(ctx, last_res, last_off)
        # the timed object's result against the one-stream pipeline's, at the timed size: the WHOLE matrix of the last step
        # reduced on the device (float64 sum + non-zero count per row, skm_matrix_row_stats; the same kernels on the same
        # values in the same order, so equal results give equal sums), then the same buffer rewritten by Pipeline.step
        ld_chk = last.shape[1]
        got_sum, got_nnz = engine.matrix_row_stats(ctx, last, n_total, n_total, ld_chk)
        probe_rows = np.unique(np.linspace(0, n_total - 1, 16).astype(np.int64))
        got_rows = [last.download(n_total, offset=int(r) * ld_chk) for r in probe_rows]
        ref_out = pipe.step(batch)
        ctx.sync()
        want_sum, want_nnz = engine.matrix_row_stats(ctx, ref_out, n_total, n_total, ld_chk)
        rows_equal = all((g == ref_out.download(n_total, offset=int(r) * ld_chk)).all() for g, r in zip(got_rows, probe_rows))
        identity = {"equal": bool((got_nnz == want_nnz).all() and (got_sum == want_sum).all() and rows_equal),
                    "rows_with_different_nonzero_count": int((got_nnz != want_nnz).sum()),
                    "rows_with_different_sum": int((got_sum != want_sum).sum()), "probe_rows_bitwise_equal": bool(rows_equal),
                    "what": "last timed-object result vs engine.Pipeline.step on the same batch: per-row float64 sums and non-zero counts "
                            "of all N x N cells (skm_matrix_row_stats) compared exactly, 16 rows compared bit for bit; parity of the "
                            "object against the oracle at this size: tests/test_gpu_parity.py::"
                            "test_config3_full_size_100k_overlapped_pipeline_the_timed_object"}
        note(f"overlapped vs single: equal={identity['equal']}")
        # host -> result for ONE batch of the stream, nothing overlapped: pack into pinned staging, asynchronous upload,
        # vectorize on the side context, cosine on the main one, wait (the PCIe-inclusive latency; never `value`)
        lat = []
        for _ in range(3):
            op.sync()
            t1 = time.perf_counter()
            op.prefetch(uploader.upload(last_res, last_off))
            op.step(None)
            op.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        stream_info["host_to_result_ms"] = min(lat)
        op.out = None
        op.close()  # (the side contexts' streams go back to the library's cache: skm_mem.hip)
        op = prof_ctxs = None
        uploader.close()
        batch = engine.SeqBatch(ctx, res, off)  # the extras below run on the first batch, resident

    shard_check = None
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if sharded:
        # After the timed region every rank checks rows of ITS block against the single-GPU path on its own device
        # (same kernels, whole batch vectorized locally, no communication): the exchange (RCCL transport included)
        # must give bit-identical rows.  64 rows spread over the block; rank 0 reports the worst rank.
        ref = engine.Pipeline(ctx, lut, args.k)
        ref.vectorize(engine.SeqBatch(ctx, res, off))
        lo_r, hi_r = bounds[rank]
        ld_r = (n_total + 3) // 4 * 4
        rows_chk = np.unique(np.linspace(lo_r, hi_r - 1, 64).astype(np.int64)) if hi_r > lo_r else np.zeros(0, np.int64)
        bad = 0
        for r in rows_chk:
            want = ref.cosine(row0=int(r), row1=int(r) + 1).download(n_total)
            got = pipe.out.download(n_total, offset=int(r - lo_r) * ld_r)
            bad += int((want != got).sum())
        del ref
        nbad = np.asarray([bad], dtype=np.int64)
        if dist is not None:
            tb = torch.tensor(nbad)
            dist.all_reduce(tb, op=dist.ReduceOp.SUM)
            nbad = tb.numpy()
        shard_check = {"rows_checked_per_rank": int(len(rows_chk)), "ranks": world, "cells_different": int(nbad[0]),
                       "what": "rows of every rank's block of the timed result against the single-GPU pipeline run on that rank "
                               "(bit for bit)"}
        note(f"sharded result check: {shard_check['cells_different']} cells differ over {world} rank(s)")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        launches, strip_ms = prof.get("k_cosine_write", (0, 0.0))
        strip_avg_ms = strip_ms / max(launches, 1)
        nnz = pipe.nnz_total if sharded else pipe.csr.nnz
        ld = (n_total + 3) // 4 * 4
        # algorithmic bytes of one k_cosine_write launch (DESIGN.md "Kernels"): the float32 output
        # rows it must write; the sparse neighbour lists it reads are <1% of that and not counted.
        # (One launch per step; written so that it stays right if a step ever splits the launch.)
        algo_bytes = rows_local * ld * 4 * args.steps / max(launches, 1)
        achieved = algo_bytes / (strip_avg_ms * 1e-3) / 1e9 if strip_avg_ms > 0 else 0.0
        traffic, traffic_note = live_traffic if live_traffic and live_traffic[0] else load_pmc_traffic()
        if live_traffic and not live_traffic[0]:
            traffic_note = f"{traffic_note} (live pass: {live_traffic[1]})"
        line = {
            "metric": BASELINE_METRIC,
            "value": n_total / (elapsed / args.steps),
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: {n_total} x {args.length}aa synthetic protein families "
                f"(100/family, 10% substitutions, shuffled), alphabet={args.alphabet} k={args.k}, "
                "recode+count(CSR)+basis+full NxN float32 cosine resident in HBM",
                "n_sequences": n_total,
                "residues": residues_total,
                "nnz": nnz,
                "basis_columns": pipe.basis.ncols,
                "parallelism": f"row-sharded x{world}, postings built by k-mer owner (RCCL all-to-all + all-gathers)"
                if sharded else "single GPU",
                "pipelining": ("engine.OverlappedPipeline fed by engine.BatchUploader: the steps of a stream of batches that ARRIVES FROM THE HOST - "
                               "three different batches cycle, one asynchronous upload per timed step from pinned staging into recycled device "
                               "buffers on a copy stream, ordered by events on the device (no hipMalloc, fill or host wait per batch).  "
                               "While a step's cosine (sparse Gram + N x N writer) "
                               "runs on the main stream, the NEXT batch's recode + count + sort + basis / postings run on a second stream "
                               "confined to half of the compute units (skm_create_confined), and so do the neighbour lists of that batch's "
                               "last 60 % of rows (skm_cosine_csr_phase), which the main stream then only has to write.  Every timed step holds one complete "
                               "vectorize and one complete cosine - the batch vectorized in the last timed step is never consumed, the one "
                               "consumed in the first was vectorized during the warm-up - and every result is recomputed from the residues, "
                               "bit-identical to the one-stream pipeline's (tests/test_gpu_parity.py).  `single_batch_step` in this line: the "
                               "same work for ONE batch on one stream (engine.Pipeline; python3 bench.py --pipeline single times that)")
                if (not sharded and args.pipeline == "overlapped") else "none: one batch at a time on one stream",
            },
            "residues_per_s": residues_total / (elapsed / args.steps),
            "stage_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
            "roofline": {
                "kernel": "k_cosine_write",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "launches": launches,
                "avg_launch_ms": strip_avg_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
                "whole_step_frac": rows_local * ld * 4 / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if stream_info is not None:
            line["stream_of_batches"] = stream_info
        if identity is not None:
            line["overlapped_equals_single"] = identity["equal"]
            line["overlapped_vs_single"] = identity
        if shard_check is not None:
            line["sharded_result_check"] = shard_check
        if sharded and pipe.mode == "distributed":
            line["stage_rooflines"] = sharded_stage_rooflines(engine, args, pipe, prof, world, int(off[hi] - off[lo]), rows_local)
            line["exchange"] = dict(pipe.sizes, collectives_per_step=4,
                                    rccl_ms_per_step={k: v[1] / args.steps for k, v in prof.items() if k.startswith("rccl_")})
        if world == 1 and not sharded and not args.no_extras:
            extras(ctx, engine, alphabet, args, line, pipe, batch, prof, res, off, seed)
        if not args.no_cpu_baseline:
            # the reference-equivalent numpy points are a rank-0, N=1 leg (3 minutes of host time); a multi-rank run
            # carries the sparse C restatement only (seconds) and says so
            pts = [int(x) for x in args.cpu_points.split(",") if x] if world == 1 else []
            note(f"CPU baseline leg (bounded to {args.cpu_budget_s:.0f} s)")
            line["cpu_baseline"] = cpu_baseline(args.alphabet, args.k, seed, pts, args.cpu_budget_s, args.cpu_sparse_n)
        note("done")
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        import faulthandler

        faulthandler.cancel_dump_traceback_later()


def sharded_stage_rooflines(engine, args, pipe, prof, world, residues_local, rows_local):
    """Rank 0's stages of one sharded step (dist.ShardedPipeline, distributed basis): algorithmic bytes / measured
    time.  Device stages are priced against HBM; the two RCCL exchanges against the per-rank xGMI egress (one link
    per peer, ~153 GB/s each: MI355X_MICROARCH.md)."""
    z = pipe.sizes
    loc, own = z["local_entries"], z["owned_entries"]
    cb = 4 if pipe.code_bits == 32 else 8
    kb = engine.key_bits(pipe.lut.nsym, pipe.k)
    passes = min((kb + 8) // 9, (kb + 7) // 8)
    n = pipe.n_total
    ld = (n + 3) // 4 * 4
    # pairs of this rank's rows: every shared entry walks its column's posting list once
    col = pipe.x.colidx.download(loc)
    cp = pipe.basis.colptr.download(z["columns"] + 1).astype(np.int64)
    ok = col != 0xFFFFFFFF
    pairs = int((cp[1:] - cp[:-1])[col[ok].astype(np.int64)].sum())
    xgmi = 153.0 * max(world - 1, 1)
    table = [
        ("k_count_short", "hbm", residues_local + loc * (cb + 4), "1 B/residue read + (code,count) per distinct k-mer written"),
        ("k_compact_rows", "hbm", loc * 2 * (cb + 4), "padded rows -> tight CSR"),
        ("k_bucket_keys", "hbm", loc * (cb + 4 + 8 + 1), "entries read; posting word + owner byte written"),
        ("rocprim_radix_sort_owner", "hbm", loc * 7, "one pass over the owner bytes (histogram read, key read + written, index written)"),
        ("k_partition_gather", "hbm", loc * (4 + 2 * (cb + 8)), "index read; code + posting word gathered and written in owner order"),
        ("k_row_norms", "hbm", loc * 4, "counts read"),
        ("rccl_alltoallv", "xgmi", z["alltoall_bytes_out"], "entries leaving this rank (code + posting word), one xGMI link per peer"),
        ("rocprim_radix_sort_owned_codes", "hbm", own * passes * 2 * (cb + 4), f"{passes} Onesweep passes x (key + index)"),
        ("rocprim_scan_shared_kmers", "hbm", own * (cb + 8), "sorted keys read, scan written"),
        ("k_bucket_emit", "hbm", own * (cb + 12) + z["owned_postings"] * 16 + z["owned_columns"] * (8 + cb),
         "sorted keys/indices/scan read; posting word gathered + written per shared entry; column start + table slot per column"),
        ("rccl_allgatherv", "xgmi", z["owned_postings"] * 8 + z["owned_columns"] * 4 + z["owned_table_slots"] * (cb + 4) + rows_local * 4,
         "this rank's own postings / column starts / table / norms, sent once over the link to every peer (bytes per link)"),
        ("k_colidx_lookup", "hbm", loc * (cb + 4) + loc * (cb + 4), "codes read, column ids written, one table probe (key + value) per entry"),
        ("k_gram_sparse", "hbm", pairs * 8, "every (row, posting) pair of this rank's rows reads one 8-byte posting"),
        ("k_cosine_write", "hbm", rows_local * ld * 4, "this rank's float32 row block"),
    ]
    out = []
    for name, bound, nbytes, what in table:
        cnt, ms = prof.get(name, (0, 0.0))
        if not cnt:
            continue
        per_step = ms / args.steps
        peak = HBM_PEAK_GBS if bound == "hbm" else (153.0 if name == "rccl_allgatherv" else xgmi)
        gbs = nbytes / (per_step * 1e-3) / 1e9 if per_step > 0 else 0.0
        out.append({"kernel": name, "bound": bound, "algorithmic_bytes_per_step": int(nbytes), "ms_per_step": per_step,
                    "achieved_GBps": gbs, "peak_GBps": peak, "frac": gbs / peak, "bytes_are": what})
    return out


def stage_rooflines(ctx, engine, args, pipe, prof, residues_total, steps=None):
    """Every stage of the step: algorithmic bytes / measured time."""
    import ctypes as C

    csr, b = pipe.csr, pipe.basis
    nnz, n = csr.nnz, csr.n
    col = csr.colidx.download(nnz)
    shared = int((col != 0xFFFFFFFF).sum())  # entries of k-mers found in >= 2 sequences
    pairs = C.c_uint64(0)
    ctx.call("skm_pair_work", C.c_int64(b.ncols), C.c_void_p(b.colptr.ptr), C.byref(pairs))
    pairs = int(pairs.value) - (nnz - shared)  # single-sequence columns were counted as one pair each
    code_b = 4 if csr.code_bits == 32 else 8
    kb = engine.key_bits(pipe.lut.nsym, pipe.k)
    passes = min((kb + 8) // 9, (kb + 7) // 8)  # 9-bit digits whenever they save a pass (skm_sort.h)
    ld = (n + 3) // 4 * 4
    fused = getattr(pipe, "fused", False)
    table = [
        ("k_count_short", residues_total + nnz * (code_b + 4), "1 B/residue read + (code,count) per distinct k-mer written"),
        ("k_compact_rows", nnz * 2 * (code_b + 4) + (nnz * 8 if fused else 0),
         "padded rows -> tight CSR: read + write every entry" + (" + the 8-byte posting word of every entry" if fused else "")),
        ("rocprim_radix_sort_codes", nnz * passes * 2 * (code_b + 4), f"{passes} Onesweep passes x (key + index payload) read + written"),
        ("k_head_count", nnz * code_b, "sorted keys read once"),
        ("k_basis_scatter", nnz * (code_b + (4 if fused else 8)) + shared * 20 + b.ncols * (code_b + 4),
         "sorted keys/indices" + ("" if fused else "/column ids") + " read; per shared entry: posting word gathered, colidx + posting "
         "written; per column: code + start"),
        ("k_gram_sparse", pairs * 8, "every (row, posting) pair reads one 8-byte posting: sum over shared columns of df^2"),
        ("k_cosine_write", n * ld * 4, "the float32 result"),
    ]
    out = []
    for name, nbytes, what in table:
        cnt, ms = prof.get(name, (0, 0.0))
        if not cnt:
            continue
        per_step = ms / (steps or args.steps)
        gbs = nbytes / (per_step * 1e-3) / 1e9
        out.append({"kernel": name, "bound": "hbm", "algorithmic_bytes_per_step": int(nbytes), "ms_per_step": per_step,
                    "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "bytes_are": what})
    return out, {"shared_entries": shared, "pairs": pairs}


def skewed_workload(ctx, engine, alphabet, args, pipe, line):
    """The same step on snekmer_amd.synth.synth_skewed (Zipf family sizes up to 5000, log-normal lengths 50-5000,
    indels, low-complexity inserts) at the benchmark's N: where the kernels tuned on uniform families of 100 fall
    off.  Parity of this generator's output: tests/test_gpu_parity.py::test_skewed_workload_20k_vs_oracle."""
    import ctypes as C

    from snekmer_amd.synth import BASE_SEED, synth_skewed

    n = args.n
    t0 = time.perf_counter()
    res, off, fam = synth_skewed(n, seed=BASE_SEED + 12)
    gen_s = time.perf_counter() - t0
    batch = engine.SeqBatch(ctx, res, off)
    p = engine.Pipeline(ctx, alphabet.build_lut(args.alphabet), args.k)
    p.out = pipe.out  # share the 40 GB result buffer
    for _ in range(2):
        p.step(batch)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 5
    per_step = []
    for _ in range(reps):  # every step timed by itself (one host wait in 24 ms): the median survives a stalled step
        t1 = time.perf_counter()
        p.step(batch)
        ctx.sync()
        per_step.append(time.perf_counter() - t1)
    dt = float(np.median(per_step))
    prof = ctx.profile_dump()
    ctx.profile_enable(False)
    st = (C.c_int64 * 4)()
    ctx.call("skm_cosine_csr_stats", st)
    lens = np.diff(off)
    sizes = np.bincount(fam)
    uniform = line.get("stage_ms_per_step", {})
    stages = {k: v[1] / reps for k, v in prof.items()}
    out = {
        "workload": f"synth_skewed({n}): family sizes Zipf(1.6) 1..{int(sizes.max())} ({len(sizes)} families, "
                    f"{int((sizes == 1).sum())} singletons), lengths {int(lens.min())}..{int(lens.max())} (median {int(np.median(lens))}), "
                    f"2 % indels, 5 % low-complexity inserts; {args.alphabet} k={args.k}",
        "residues": int(off[-1]), "nnz": p.csr.nnz, "basis_columns": p.basis.ncols, "generator_s": gen_s,
        "ms_per_step": dt * 1e3, "ms_per_step_each": [round(x * 1e3, 3) for x in per_step], "sequences_per_s": n / dt,
        "residues_per_s": int(off[-1]) / dt,
        "rows_sent_to_large_table_pass": int(st[0]), "strips_left_to_cursor_kernel": int(st[1]),
        "neighbour_list_entries_behind_fixed_slots": int(st[2]), "wide_strips": int(st[3]),
        "stage_ms_per_step": stages,
        "stage_vs_uniform_families": {k: (v / uniform[k] if uniform.get(k) else None) for k, v in stages.items()},
    }
    # the heavy rows' long-list columns on the matrix cores (skm_heavy_panel.h): chosen by the library from the
    # previous step's heavy-row count, so the warm-up steps above switched it on
    ps = (C.c_int64 * 6)()
    ctx.call("skm_heavy_panel_stats", ps)
    gemm_ms = stages.get("k_panel_gemm", 0.0)
    out["heavy_panels"] = {
        "used": bool(gemm_ms > 0), "heavy_rows": int(ps[0]), "blocks_of_256_rows": int(ps[1]), "blocks_without_panel": int(ps[2]),
        "mean_panel_columns": ps[3] / max(ps[1] - ps[2], 1), "mean_panel_rows": ps[4] / max(ps[1] - ps[2], 1),
        "gemm_int8_ops": int(2 * ps[5]), "gemm_ms": gemm_ms,
        "mfma_util": (2 * ps[5] / (gemm_ms * 1e-3) / (I8_PEAK_TOPS * 1e12)) if gemm_ms > 0 else None,
        "mfma_util_note": "k_panel_gemm builds every B tile in LDS from the posting lists (a cursor per column, one thread each) "
                          "before multiplying: the kernel's time is that fill, not the matrix pipe",
        "pipeline_ms": sum(v for kk, v in stages.items() if kk.startswith("k_panel_") or kk == "onesweep_sort_heavy_rows"),
    }
    # the same batch with the panels off (every posting list walked): what they buy
    from snekmer_amd import _hip

    _hip.set_option("SKM_HEAVY_PANEL", 0)
    try:
        p.step(batch)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            p.step(batch)
        ctx.sync()
        out["ms_per_step_without_panels"] = (time.perf_counter() - t1) / 3 * 1e3
    finally:
        _hip.set_option("SKM_HEAVY_PANEL", None)
    p.out = None
    return out


def real_proteome(ctx, engine, alphabet):
    """Real sequences: the proteome the reference's CI runs (UP000322080, 3 383 proteins, lengths 30-2 900; a data
    fixture under tests/golden/data), vectorize + full N x N cosine at the CI's own configuration (solvacc k=8: a
    3^8 = 6 561-column basis, every row shares k-mers with every other: the dense regime) and at k=12 in a
    reference alphabet (standard: the sparse regime).  Parity of both: tests/test_gpu_parity.py::test_real_proteome_*."""
    import ctypes as C

    from snekmer_amd.io import read_fasta_packed

    path = os.path.join(ROOT, "tests", "golden", "data", "UP000322080_2603819.fasta")
    if not os.path.exists(path):
        return {"skipped": "fixture not found"}
    ids, res, off = read_fasta_packed(path)
    batch = engine.SeqBatch(ctx, res, off)
    out = {"file": "tests/golden/data/UP000322080_2603819.fasta", "sequences": int(len(ids)), "residues": int(off[-1]),
           "longest": int(np.diff(off).max()), "runs": []}
    # solvacc k=8 twice: the product's own dispatch (small full basis -> int8 GEMM on the matrix cores) and the sparse
    # kernels forced on the same batch (what every round before this one ran)
    for name, k, dense_route in (("solvacc", 8, "auto"), ("solvacc", 8, False), ("standard", 12, "auto")):
        p = engine.Pipeline(ctx, alphabet.build_lut(name), k, dense_route=dense_route)
        for _ in range(3):
            p.step(batch)
        ctx.sync()
        reps = 50
        t1 = time.perf_counter()
        for _ in range(reps):
            p.step(batch)
        ctx.sync()
        dt = (time.perf_counter() - t1) / reps
        run = {"alphabet": name, "k": k, "route": p.route, "ms_per_step": dt * 1e3, "sequences_per_s": len(ids) / dt,
               "residues_per_s": int(off[-1]) / dt}
        if p.route == "dense":
            kdim = (p.lut.nsym**k + 127) // 128 * 128
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(10):
                p.step(batch)
            gemm_ms = ctx.profile_dump()["k_cosine_dense_i8"][1] / 10
            ctx.profile_enable(False)
            ops = len(ids) * (len(ids) + 1) * kdim  # symmetric launch: tiles on or above the diagonal
            run.update({"irregular_rows_recomputed_exactly": p.irregular_rows(), "gemm_ms": gemm_ms,
                        "mfma_util": ops / (gemm_ms * 1e-3) / (I8_PEAK_TOPS * 1e12), "kdim": kdim})
        else:
            st = (C.c_int64 * 4)()
            ctx.call("skm_cosine_csr_stats", st)
            run.update({"rows_handed_to_k_cosine_heavy": int(st[0]), "strips_left_to_cursor_kernel": int(st[1])})
        run.update({"nnz": p.csr.nnz, "basis_columns": p.basis.ncols})
        out["runs"].append(run)
        del p
    return out


def small_batches(ctx, engine, alphabet, args):
    """The launch-bound regime: BASELINE configs[1] (10 k sequences) and the reference's real job sizes (one FASTA file
    of 50-3 700 records per Snakemake job, snekmer/rules/kmerize.smk:57-65).  Per N: wall time per step (host running
    ahead, no synchronisation inside the loop), the same with whole steps replayed as HIP graphs (engine.Pipeline(graphs=
    "auto")), the sum of the per-stage HIP-event times, `gpu_operations_per_step` = every kernel, fill and copy of a step
    (the node count of the step captured as a HIP graph: what rocprofv3 lists as dispatches), the profiled scopes per step
    (a library sort counts as one) and `frac` = the step's HBM floor (4 bytes per result cell at the 8 TB/s spec) over the wall time."""
    from snekmer_amd.synth import BASE_SEED, synth_families

    lut = alphabet.build_lut(args.alphabet)
    out = []
    for n in (1000, 3383, 10000):
        res, off, _ = synth_families(n, args.length, family=100, seed=BASE_SEED + 1)
        b = engine.SeqBatch(ctx, res, off)
        p = engine.Pipeline(ctx, lut, args.k)
        walls = {}
        for mode in (False, "auto"):
            p.graphs = mode
            for _ in range(5):
                p.step(b)
            ctx.sync()
            reps = 200
            t1 = time.perf_counter()
            for _ in range(reps):
                p.step(b)
            ctx.sync()
            walls[mode] = (time.perf_counter() - t1) / reps * 1e3
        nodes = [ent["graph"].nodes for ent in p._graphs.values() if ent["graph"] is not None]
        p.graphs = False
        p.drop_graphs()
        wall = walls[False]
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(50):
            p.step(b)
        prof = ctx.profile_dump()
        ctx.profile_enable(False)
        ld = (n + 3) // 4 * 4
        floor_ms = 4.0 * n * ld / (HBM_PEAK_GBS * 1e9) * 1e3
        out.append({"n": n, "wall_ms": wall, "wall_ms_graph_replay": walls["auto"], "kernel_sum_ms": sum(v[1] for v in prof.values()) / 50,
                    "gpu_operations_per_step": nodes[0] if nodes else None,
                    "profiled_scopes_per_step": sum(v[0] for v in prof.values()) / 50, "hbm_floor_ms": floor_ms, "frac": floor_ms / wall,
                    "sequences_per_s": n / (wall * 1e-3), "stage_ms": {kk: round(v[1] / 50, 4) for kk, v in prof.items()}})
        del p, b
    return out


def api_vectorize_fasta(args, seed):
    """What a Snekmer user sees: kmerize.vectorize_fasta (the body of rules/kmerize.smk:67-142) on a synthetic FASTA
    FILE, host time included, split into parse (threaded C reader -> packed residues), device work that produces
    integers (H2D, counts, basis, first-seen order, count matrix in kmerlist order, D2H), string formatting on the
    device (reduced sequences and the k-mer strings of the basis as numpy '<U' arrays, D2H) and the compressed
    .npz write.  SURVEY 8(d): 'parse time reported separately'."""
    import tempfile

    from snekmer_amd import kmerize
    from snekmer_amd.synth import synth_families, to_records, write_fasta

    out = {"what": "snekmer_amd.kmerize.vectorize_fasta(path, alphabet, k): FASTA file -> kmerlist / ids / seqs / lengths / "
                   "CSR counts in kmerlist order (the sparse .npz contract); second call of two", "runs": []}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for n in (10000, 100000):
            res, off, _ = synth_families(n, args.length, family=100, seed=seed)
            path = os.path.join(tmp, f"synth_{n}.fasta")
            write_fasta(path, to_records(res, off))
            rec = None
            for rep in range(2):
                tm = {}
                t0 = time.perf_counter()
                r = kmerize.vectorize_fasta(path, args.alphabet, args.k, dense=False, timings=tm)
                wall = time.perf_counter() - t0
                rec = {"n": n, "file_bytes": os.path.getsize(path), "basis_columns": int(len(r["kmerlist"])),
                       "entries": int(len(r["counts_val"])), "parse_s": tm.get("parse_s", 0.0), "pack_s": 0.0,
                       "gpu_s": tm.get("gpu_s", 0.0), "decode_s": tm.get("decode_s", 0.0), "wall_s_without_write": wall,
                       "sequences_per_s": n / wall, "residues_per_s": int(off[-1]) / wall,
                       "result_bytes": int(sum(v.nbytes for v in r.values()))}
                del r
            tm = {}
            t0 = time.perf_counter()
            kmerize.vectorize_fasta(path, args.alphabet, args.k, sparse_npz_out=os.path.join(tmp, f"out_{n}.npz"), timings=tm)
            rec["write_s"] = tm.get("write_s", 0.0)
            rec["npz_bytes"] = os.path.getsize(os.path.join(tmp, f"out_{n}.npz"))
            rec["sequences_per_s_with_write"] = n / (time.perf_counter() - t0)
            out["runs"].append(rec)
            note(f"api_vectorize_fasta n={n}: {rec['sequences_per_s']:.0f} seq/s without the write "
                 f"(parse {rec['parse_s']:.3f} gpu {rec['gpu_s']:.3f} decode {rec['decode_s']:.3f} s), write {rec['write_s']:.1f} s")
    return out


def apply_chain(ctx, engine, alphabet, args, seed):
    """SURVEY 8(f) rows f1/f2, the reference's only real consumer of the cosine (rules/apply.smk:278-342, rules/learn.smk:385-408,
    811-849): N query sequences against A family-total rows.  Timed: the learn aggregation (skm_csr_group_sum), the fused
    epilogue (skm_apply_top2: exact int64 dots, float64 scores, top-2 per query, the N x A block never stored) and, for
    comparison, the materialised N x A float32 block (skm_cosine_csr) + skm_row_top2.  Roofline of k_apply_top2: postings
    walked x 8 B (every entry of a query row reads the posting list of its column in the totals matrix)."""
    from snekmer_amd import apply as skm_apply
    from snekmer_amd.synth import synth_families

    n = args.n
    lut = alphabet.build_lut(args.alphabet)
    res, off, fam = synth_families(n, args.length, family=100, seed=seed)
    batch = engine.SeqBatch(ctx, res, off)
    csr = engine.count_csr(ctx, batch, lut, args.k)
    # the count matrix by column as well (the postings every vectorize step builds for the cosine): the learn aggregation
    # sums list by list there and never sorts (skm_group_postings)
    basis = engine.build_basis(ctx, csr, lut.nsym, args.k, postings=True)
    nfam = int(fam.max()) + 1
    groups = fam.astype(np.uint32)
    totals = skm_apply.group_sum(ctx, csr, groups, nfam, basis=basis)
    # postings walked: sum over query entries of the column's document frequency in the totals matrix
    tcol = totals.colidx.download(totals.nnz)
    df = np.bincount(tcol, minlength=basis.ncols)
    walked = int(df[csr.colidx.download(csr.nnz)].sum())
    out = {"workload": f"{n} queries x {nfam} family totals ({args.alphabet} k={args.k}; the families of the bench workload, 100 members each)",
           "queries": n, "families": nfam, "query_entries": csr.nnz, "totals_entries": totals.nnz, "basis_columns": basis.ncols,
           "postings_walked": walked}
    reps = 5

    def timed(fn):
        fn()
        ctx.sync()
        ctx.profile_enable(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        prof = {k: v[1] / reps for k, v in ctx.profile_dump().items()}
        ctx.profile_enable(False)
        return wall, prof, r

    w, pr, _ = timed(lambda: skm_apply.group_sum(ctx, csr, groups, nfam, basis=basis))
    out["group_sum"] = {"ms_incl_host": w, "kernel_ms": {k: round(v, 4) for k, v in pr.items() if v > 0.002}, "device_ms": sum(pr.values()),
                        "what": "totals by column from the count matrix's postings (skm_group_postings: what skm_apply_top2 reads) AND the "
                                "[families x columns] CSR of learn.smk:385-408 from them (skm_postings_to_csr, the library's own one-sweep by family)"}
    w, pr, _ = timed(lambda: skm_apply.group_totals(ctx, csr, groups, nfam, basis=basis))
    gp_bytes = csr.nnz * 8 + basis.ncols * 12 + totals.nnz * 24
    out["group_totals_by_column"] = {"ms_incl_host": w, "kernel_ms": {k: round(v, 4) for k, v in pr.items() if v > 0.002}, "device_ms": sum(pr.values()),
                                     "roofline": {"bound": "hbm", "algorithmic_bytes": gp_bytes,
                                                  "bytes_are": "8 B per posting read + 12 B per column (two column starts read, one written) + 24 B per total (written to the "
                                                               "column's slot, read and written once more by the packing pass)",
                                                  "frac": gp_bytes / (sum(pr.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS if sum(pr.values()) > 0 else None}}
    w, pr, _ = timed(lambda: skm_apply.group_sum(ctx, csr, groups, nfam))
    out["group_sum_from_csr_only"] = {"ms_incl_host": w, "kernel_ms": {k: round(v, 4) for k, v in pr.items() if v > 0.002}, "device_ms": sum(pr.values()),
                                      "what": "no postings given: the CSR is transposed first (one 24-bit vendor radix sort of all entries)"}
    w, pr, r = timed(lambda: skm_apply.apply_top2(ctx, csr, basis.ncols, totals))
    k_ms = pr.get("k_apply_top2", 0.0)
    dev_ms = sum(pr.values())
    out["fused_apply_top2"] = {
        "ms_incl_host": w, "device_ms": dev_ms, "kernel_ms": {k: round(v, 4) for k, v in pr.items() if v > 0.002},
        "sequences_per_s_device": n / (dev_ms * 1e-3) if dev_ms > 0 else None,
        "roofline": {"kernel": "k_apply_top2", "bound": "hbm", "algorithmic_bytes": walked * 8 + csr.nnz * 16 + n * 40,
                     "bytes_are": "8 B per posting walked + 16 B per query entry (column id, count, column start and end) + 40 B per row written",
                     "ms": k_ms, "achieved_GBps": (walked * 8 + csr.nnz * 16 + n * 40) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                     "frac": (walked * 8 + csr.nnz * 16 + n * 40) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                     "note": "wave per query row; one gather per entry (the column's word: family and total when one family holds the k-mer), "
                             "sums in a per-wave LDS hash table keyed by the family; rows touching more than 384 families go to the dense form"}}
    out["top1_is_own_family_frac"] = float(np.mean(r[0][:, 0] == fam))
    # the same call with the rows PROCESSED in label order (learn.smk's self-evaluation knows the labels; real inputs arrive
    # one family per FASTA file): the kernel's one random 128-byte line per query entry then hits in L2.  Same results.
    by_label = np.argsort(fam, kind="stable").astype(np.uint32)
    w2, pr2, r2o = timed(lambda: skm_apply.apply_top2(ctx, csr, basis.ncols, totals, order=by_label))
    out["fused_apply_top2_rows_in_label_order"] = {
        "ms_incl_host": w2, "device_ms": sum(pr2.values()), "kernel_ms": {k: round(v, 4) for k, v in pr2.items() if v > 0.002},
        "equal_to_the_shuffled_order": bool((r2o[0] == r[0]).all() and (r2o[1] == r[1]).all() and (r2o[2] == r[2]).all()),
        "k_apply_top2_frac": (walked * 8 + csr.nnz * 16 + n * 40) / (pr2.get("k_apply_top2", 0.0) * 1e-3) / 1e9 / HBM_PEAK_GBS if pr2.get("k_apply_top2") else None}

    def unfused():
        S, ld = skm_apply.cosine_rows_vs_totals(ctx, csr, basis.ncols, totals)
        return skm_apply.row_top2(ctx, S, n, nfam, ld)

    w, pr, r2 = timed(unfused)
    out["materialised_block_then_top2"] = {"ms_incl_host": w, "device_ms": sum(pr.values()), "block_bytes": int(n * ((nfam + 3) // 4 * 4) * 4),
                                           "kernel_ms": {k: round(v, 4) for k, v in pr.items() if v > 0.002}}
    out["fused_equals_materialised_top1"] = bool((r2[0][:, 0] == r[0][:, 0]).all())
    out["parity"] = "tests/test_gpu_parity.py::test_apply_epilogue_matches_reference_rule_golden (delta / Confidence equal to G12)"
    return out


def config4_one_rank_share(ctx, engine, alphabet, args):
    """BASELINE configs[3] (1 M x 300 aa sharded 8 ways) as far as ONE GPU can measure it: all 1 M sequences vectorized here
    (a rank of the sharded job vectorizes 1/8 of them and receives the postings of the rest), then one rank's share of the
    pairwise step: exact neighbour lists of 125 k rows against all 1 M columns and the top-10 per row (the 1 M x 1 M float32
    matrix would be 4 TB; DESIGN.md section 2, decision 2).  The 8-GPU figure derived from it is an ESTIMATE, labelled so."""
    from snekmer_amd.synth import BASE_SEED, synth_families

    n, world = 1_000_000, 8
    block = n // world
    lut = alphabet.build_lut(args.alphabet)
    t0 = time.perf_counter()
    res, off, fam = synth_families(n, args.length, family=100, seed=BASE_SEED + 3)
    gen_s = time.perf_counter() - t0
    batch = engine.SeqBatch(ctx, res, off)
    pipe = engine.Pipeline(ctx, lut, args.k)
    rec = None
    for rnd in range(2):
        ctx.profile_enable(True)
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        pipe.vectorize(batch)
        ctx.sync()
        t_vec = time.perf_counter() - t0
        b = pipe.basis
        t0 = time.perf_counter()
        nb = engine.gram_neighbors(ctx, pipe.csr, pipe.rnorm, n, b.ncols, b.colptr, b.post, pipe.rnorm, row0=0, row1=block,
                                   cap_entries=block * 6000, post_bits=b.post_bits, postcnt=b.postcnt)
        ctx.sync()
        t_nb = time.perf_counter() - t0
        t0 = time.perf_counter()
        idx, val = engine.neighbors_topk(ctx, nb, pipe.rnorm, pipe.rnorm, 10, exclude_self=True)
        t_top = time.perf_counter() - t0
        prof = ctx.profile_dump()
        ctx.profile_enable(False)
        rec = (t_vec, t_nb, t_top, prof, nb.total, nb.overflow_rows)
        if rnd == 0:
            del nb
    t_vec, t_nb, t_top, prof, entries, ovf = rec
    same = float(np.mean(fam[idx[:, 0].astype(np.int64) % n] == fam[:block]))
    # rooflines of this configuration's own kernels (the 1 M-row forms: other kernels than config 3's step)
    import ctypes as C

    pairs = C.c_uint64(0)
    ctx.call("skm_pair_work", C.c_int64(b.ncols), C.c_void_p(b.colptr.ptr), C.byref(pairs))
    nnz = pipe.csr.nnz
    code_b = 4 if pipe.csr.code_bits == 32 else 8
    shared = int(b.colptr.download(1, offset=b.ncols)[0])  # postings = entries of k-mers found in >= 2 sequences
    block_pairs = int(pairs.value) // world  # the rows are shuffled: a row block holds 1/8 of the matrix's (row, posting) pairs
    rooflines = []
    for names, nbytes, what in (
            (("k_gram_sparse", "k_gram_sparse_big", "k_gram_sparse_huge"), block_pairs * 8,
             "every (row, posting) pair of the 125 k-row block reads one 8-byte posting: 1/8 of the sum over shared columns of df^2"),
            (("k_neighbors_topk", "k_neighbors_topk_lds", "k_neighbors_topk_stream"), int(entries) * 8 + block * 10 * 8,
             "8 B per list entry read + 10 x 8 B (index, score) written per row"),
            (("k_basis_scatter", "k_basis_scatter_fused"), nnz * (code_b + 4) + shared * 20 + b.ncols * (code_b + 4),
             "1 M rows: sorted keys / indices read; per shared entry: posting word gathered, colidx + posting written; per column: code + start")):
        ms = sum(prof[k][1] for k in names if k in prof)
        if ms > 0:
            gbs = nbytes / (ms * 1e-3) / 1e9
            rooflines.append({"kernels": [k for k in names if k in prof], "bound": "hbm", "algorithmic_bytes": int(nbytes), "ms": ms,
                              "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "bytes_are": what})
    # a rank of the 8-GPU job: vectorize of its 125 k sequences (1/8 of the vectorize measured here), the exchange (not
    # measurable on one GPU: DESIGN.md section 7 prices it), then exactly the pairwise share measured here
    est = t_vec / world + t_nb + t_top
    return {
        "workload": f"BASELINE configs[3]: {n} x {args.length}aa, {args.alphabet} k={args.k}; one rank's share on one MI355X",
        "generator_s": gen_s, "nnz": pipe.csr.nnz, "basis_columns": b.ncols,
        "vectorize_1m_ms": t_vec * 1e3, "vectorize_sequences_per_s": n / t_vec,
        "neighbour_lists_125k_x_1m_ms": t_nb * 1e3, "list_entries": int(entries), "entries_per_row": entries / block, "overflow_rows": int(ovf),
        "top10_ms_incl_download": t_top * 1e3, "top1_same_family_frac": same,
        "kernel_ms": {k: round(v[1], 3) for k, v in prof.items() if v[1] > 0.05},
        "stage_rooflines": rooflines,
        "ESTIMATE_8_gpu": {"ms_per_job": est * 1e3, "sequences_per_s": n / est,
                           "how": "vectorize_1m_ms / 8 + neighbour_lists_125k_x_1m_ms + top10_ms: every rank's device work if the postings "
                                  "exchange (0.38 GB out, 2.2 GB in per rank over xGMI) were free; an estimate from one GPU, NOT a measurement "
                                  "of 8 GPUs"},
        "parity": "tests/test_gpu_parity.py::test_config4_one_rank_share_125k_rows_vs_1m, tests/test_gpu_sharded.py (8 thread-ranks, 1 M sequences)",
    }


def extras(ctx, engine, alphabet, args, line, pipe, batch, prof, res, off, seed):
    """Measurements taken after the timed region, on the same GPU in the same run."""
    import ctypes as C

    from snekmer_amd.synth import BASE_SEED, synth_families

    n_total = args.n
    # ONE batch on one stream (engine.Pipeline), nothing of another batch beside it: the single-batch latency, and the
    # stage times the per-stage rooflines are taken from (in the pipelined timed region every stage shares the chip with
    # another batch's kernels)
    pipe.step(batch)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t1 = time.perf_counter()
    for _ in range(10):
        pipe.step(batch)
    ctx.sync()
    one_ms = (time.perf_counter() - t1) / 10 * 1e3
    one_prof = ctx.profile_dump()
    ctx.profile_enable(False)
    w_calls, w_ms = one_prof.get("k_cosine_write", (0, 0.0))
    w_avg = w_ms / max(w_calls, 1)
    w_bytes = n_total * ((n_total + 3) // 4 * 4) * 4
    note("extras: single_batch_step")
    line["single_batch_step"] = {
        "what": "engine.Pipeline.step: one batch, every kernel back to back on one stream (python3 bench.py --pipeline single times this as "
                "the headline)",
        "ms_per_step": one_ms, "sequences_per_s": n_total / (one_ms * 1e-3),
        "stage_ms_per_step": {k: v[1] / 10 for k, v in one_prof.items()},
        "k_cosine_write_alone": {"avg_launch_ms": w_avg, "GBps": w_bytes / (w_avg * 1e-3) / 1e9 if w_avg > 0 else None,
                                 "frac_of_hbm_peak": w_bytes / (w_avg * 1e-3) / 1e9 / HBM_PEAK_GBS if w_avg > 0 else None},
    }
    line["roofline"]["single_batch"] = {"avg_launch_ms": w_avg, "achieved": w_bytes / (w_avg * 1e-3) / 1e9 if w_avg > 0 else None,
                                        "frac": w_bytes / (w_avg * 1e-3) / 1e9 / HBM_PEAK_GBS if w_avg > 0 else None,
                                        "what": "the same kernel in single_batch_step: one launch with the chip to itself"}
    line["stage_rooflines"], counts = stage_rooflines(ctx, engine, args, pipe, one_prof, int(off[-1]), steps=10)
    line["config"].update(counts)

    # host -> result: H2D of the packed batch + one step, synchronised (FASTA parse excluded)
    times = []
    for _ in range(3):
        ctx.sync()
        t1 = time.perf_counter()
        b2 = engine.SeqBatch(ctx, res, off)
        pipe.step(b2)
        ctx.sync()
        times.append((time.perf_counter() - t1) * 1e3)
        del b2
    note("extras: host_to_result_ms")
    line["host_to_result_ms"] = {"value": min(times), "what": "H2D of residues + offsets (pageable host memory) + one step, result left in HBM",
                                 "h2d_bytes": int(res.nbytes + off.nbytes)}

    # the opt-in two-stream schedule of skm_cosine_csr (neighbour lists of block b+1 built while block b is written):
    # faster as a whole, but both kernels slow each other down, so the per-kernel rooflines above are taken without it
    from snekmer_amd import _hip

    _hip.set_option("SKM_COSINE_OVERLAP", 1)
    try:
        pipe.step(batch)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            pipe.step(batch)
        ctx.sync()
        ov_ms = (time.perf_counter() - t1) / 5 * 1e3
    finally:
        _hip.set_option("SKM_COSINE_OVERLAP", None)
    note("extras: overlap_schedule")
    line["overlap_schedule"] = {"ms_per_step": ov_ms, "sequences_per_s": n_total / (ov_ms * 1e-3),
                                "what": "SKM_COSINE_OVERLAP=1: same step in 8 row blocks, the Gram of block b+1 on a stream confined to half of the "
                                        "compute units beside the writer of block b on an unconfined stream"}

    if args.pipeline != "overlapped":  # (the timed region itself otherwise)
        # engine.OverlappedPipeline: a stream of batches, batch i+1 vectorized on a second context while batch i's cosine runs
        op = engine.OverlappedPipeline(ctx, pipe.lut, pipe.k)
        op.out = pipe.out  # share the 40 GB result buffer
        op.prefetch(batch)
        op.step(batch)
        op.sync()
        t1 = time.perf_counter()
        for _ in range(10):
            op.step(batch)
        op.sync()
        op_ms = (time.perf_counter() - t1) / 10 * 1e3
        op.step(None)  # consume the batch still prefetched
        op.sync()
        op.out = None
        del op
        note("extras: overlapped_pipeline")
        line["overlapped_pipeline"] = {"ms_per_step": op_ms, "sequences_per_s": n_total / (op_ms * 1e-3),
                                       "what": "engine.OverlappedPipeline, a stream of batches: the next batch's count / sort / scatter run on a "
                                               "second context (own stream confined to half of the compute units by skm_create_confined, second "
                                               "buffer set) beside this batch's cosine; every step holds one complete vectorize and one complete "
                                               "cosine; results identical to Pipeline's"}

    if args.alphabet == "red6":
        # red6 is a benchmark alphabet; the nearest reference alphabet (`standard`,
        # 7^12 needs uint64 codes) is timed next to it on the same sequences
        lut7 = alphabet.build_lut("standard")
        pipe7 = engine.Pipeline(ctx, lut7, args.k)
        pipe7.out = pipe.out  # share the 40 GB result buffer
        pipe7.step(batch)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            pipe7.step(batch)
        ctx.sync()
        dt7 = (time.perf_counter() - t1) / 3
        note("extras: reference_alphabet_check")
        line["reference_alphabet_check"] = {
            "alphabet": "standard", "k": args.k, "code_bits": 64, "ms_per_step": dt7 * 1e3,
            "sequences_per_s": n_total / dt7, "nnz": pipe7.csr.nnz, "basis_columns": pipe7.basis.ncols,
        }
        pipe7.out = None
        del pipe7

    line["skewed_workload"] = skewed_workload(ctx, engine, alphabet, args, pipe, line)
    note("extras: skewed_workload")
    line["real_proteome"] = real_proteome(ctx, engine, alphabet)
    note("extras: real_proteome")
    line["api_vectorize_fasta"] = api_vectorize_fasta(args, seed)
    note("extras: api_vectorize_fasta")
    line["small_batches"] = small_batches(ctx, engine, alphabet, args)
    note("extras: small_batches")

    # BASELINE configs[1]: 10k sequences, same alphabet and k
    res2, off2, _ = synth_families(10000, args.length, family=100, seed=BASE_SEED + 1)
    b2 = engine.SeqBatch(ctx, res2, off2)
    p2 = engine.Pipeline(ctx, alphabet.build_lut(args.alphabet), args.k)
    for _ in range(3):
        p2.step(b2)
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(50):
        p2.step(b2)
    ctx.sync()
    dt2 = (time.perf_counter() - t1) / 50
    note("extras: config2")
    line["config2"] = {"workload": f"BASELINE configs[1]: 10000 x {args.length}aa, {args.alphabet} k={args.k}",
                       "ms_per_step": dt2 * 1e3, "sequences_per_s": 10000 / dt2}
    del p2, b2

    # the 40 GB result is no longer needed: make room for the 210 GB count matrix
    pipe.out = None
    _, _, mem = ctx.device_info()

    line["apply_chain"] = apply_chain(ctx, engine, alphabet, args, seed)
    note("extras: apply_chain")
    if mem >= 150 * 2**30:
        line["config4_one_rank_share"] = config4_one_rank_share(ctx, engine, alphabet, args)
        note("extras: config4_one_rank_share")

    # BASELINE configs[4]: dense count scatter, hydro k=20, uint16 cells
    lut2 = alphabet.build_lut("hydro")
    n5 = n_total if mem >= 250 * 2**30 else 20000
    res5, off5, _ = synth_families(n5, args.length, family=100, seed=BASE_SEED + 4)
    b5 = engine.SeqBatch(ctx, res5, off5)
    dense = engine.count_dense(ctx, b5, lut2, 20, dtype=np.uint16)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 2
    for _ in range(reps):
        engine.count_dense(ctx, b5, lut2, 20, dtype=np.uint16, out=dense)
    p5 = ctx.profile_dump()
    ctx.profile_enable(False)
    windows = int(np.maximum(np.diff(off5) - 20 + 1, 0).sum())
    cells = n5 * dense.shape[1]
    ms_fill = p5["memset_count_dense"][1] / reps
    ms_sc = p5["k_count_dense"][1] / reps
    bytes5 = int(res5.size + cells * 2 + windows * 4)  # N*L + N*B*2 (zero fill) + N*W*2*2 (RMW)
    note("extras: config5_count_dense")
    line["config5_count_dense"] = {
        "workload": f"BASELINE configs[4]: {n5} x {args.length}aa, hydro k=20, dense uint16 [{n5} x {dense.shape[1]}]",
        "matrix_bytes": int(cells * 2), "windows": windows, "memset_ms": ms_fill, "k_count_dense_ms": ms_sc,
        "ms": ms_fill + ms_sc, "algorithmic_bytes": bytes5, "TBps": bytes5 / (ms_fill + ms_sc) / 1e9,
        "frac": bytes5 / ((ms_fill + ms_sc) * 1e-3) / 1e9 / HBM_PEAK_GBS, "atomics_per_s": windows / ms_sc * 1e3,
        "sequences_per_s": n5 / ((ms_fill + ms_sc) * 1e-3),
        "parity": "tests/test_gpu_parity.py::test_config5_count_dense_full_size_100k_by_2pow20 (every cell vs the oracle)",
    }
    del dense, b5

    # the MFMA cosine: hydro k=14 (16384 columns), N = 32768
    nm = 32768
    resm, offm, _ = synth_families(nm, args.length, family=100, seed=BASE_SEED + 6)
    bm = engine.SeqBatch(ctx, resm, offm)
    dp = engine.DensePipeline(ctx, lut2, 14)
    dp.step(bm)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t1 = time.perf_counter()
    for _ in range(3):
        dp.step(bm)
    ctx.sync()
    dt_pipe = (time.perf_counter() - t1) / 3
    pm = ctx.profile_dump()
    ms_sym = pm["k_cosine_dense_i8"][1] / pm["k_cosine_dense_i8"][0]
    # the same kernel without the X-is-Y shortcut (what a rectangular X, Y call runs): every tile computed
    from snekmer_amd import _hip

    _hip.set_option("SKM_DENSE_VARIANT", 11)
    ctx.profile_reset()
    for _ in range(3):
        engine.cosine_dense_i8(ctx, nm, nm, dp.kdim, dp.dense, dp.dense, dp.rnorm, dp.rnorm, out=dp.out)
    ms_full = ctx.profile_read("k_cosine_dense_i8")[1] / 3
    _hip.set_option("SKM_DENSE_VARIANT", None)
    ctx.profile_enable(False)
    ops_full = 2.0 * nm * nm * dp.kdim
    nt = -(-nm // 256)
    ops_sym = (nt * (nt + 1) // 2) * 2.0 * 256 * 256 * dp.kdim  # MFMA work the symmetric launch executes
    note("extras: dense_mfma")
    line["dense_mfma"] = {
        "shape": f"N = M = {nm}, K = {dp.kdim} (hydro k=14 full basis), int8 x int8 -> int32 -> float32",
        "kernel": "k_cosine_dense_i8_v5 (v_mfma_i32_32x32x32_i8, 256x256 tile, 1 x 8 waves in two staggered groups; A staged through "
                  "LDS from a tiled copy, B loaded straight into the MFMA operand registers from a lane-order copy; both copies "
                  "are made inside the timed call)",
        "ms": ms_full, "ops": ops_full, "POPS": ops_full / (ms_full * 1e-3) / 1e15, "peak_POPS": I8_PEAK_TOPS / 1e3,
        "frac": ops_full / (ms_full * 1e-3) / 1e12 / I8_PEAK_TOPS,
        "symmetric": {"what": "X is Y: tiles on or above the diagonal only, mirrored stores (what DensePipeline runs)",
                      "ms": ms_sym, "executed_ops": ops_sym, "POPS_executed": ops_sym / (ms_sym * 1e-3) / 1e15,
                      "frac_executed": ops_sym / (ms_sym * 1e-3) / 1e12 / I8_PEAK_TOPS,
                      "speedup_over_full": ms_full / ms_sym},
        "bound": "L2 -> CU operand traffic: 32 KiB per tile and K stage = 137 GB per launch; every variant measured in round 4 "
                 "(v5, v4 with B in registers, a software-pipelined loop) stops at 137 GB / 11.2 ms = 12.2 TB/s, the rate round "
                 "2 measured for the staging alone (the microarchitecture guide: 17-19 TB/s for L2-resident rows, 7-9 TB/s "
                 "beyond L2; 88 % of the requests hit); profiles/r04_dense_mfma.json",
        "dense_pipeline_ms_per_step": dt_pipe * 1e3, "dense_pipeline_sequences_per_s": nm / dt_pipe,
        "stage_ms": {k: v[1] / v[0] for k, v in pm.items()},
    }


if __name__ == "__main__":
    main()
