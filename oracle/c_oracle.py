"""ORACLE (test infrastructure): ctypes wrapper over oracle/kmer_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    path = os.path.join(_HERE, "_build", "libkmer_oracle.so")
    src = os.path.join(_HERE, "kmer_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return path


def lib():
    global _LIB
    if _LIB is None:
        # SKM_ORACLE_LIB: another build of the same source (the -fsanitize build of `make -C oracle asan`)
        _LIB = C.CDLL(os.environ.get("SKM_ORACLE_LIB") or build())
        _LIB.orc_count_csr.restype = C.c_int64
        _LIB.orc_basis.restype = C.c_int64
        _LIB.orc_count_csr_mt.restype = C.c_int64
        _LIB.orc_basis_mt.restype = C.c_int64
        _LIB.orc_cosine_all_mt.restype = C.c_double
        _LIB.orc_max_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def recode(table, seq, off):
    n = len(off) - 1
    out = np.zeros_like(seq)
    outlen = np.zeros(n, dtype=np.int32)
    lib().orc_recode(_p(table), _p(seq), _p(off), C.c_int64(n), _p(out), _p(outlen))
    return out, outlen


def kmer_codes(rank, nsym, k, seq, off):
    n = len(off) - 1
    codes = np.full(len(seq) + 1, np.iinfo(np.uint64).max, dtype=np.uint64)
    nwin = np.zeros(n, dtype=np.int32)
    lib().orc_kmer_codes(_p(rank), nsym, k, _p(seq), _p(off), C.c_int64(n), _p(codes), _p(nwin))
    return codes[: len(seq)], nwin


def host_threads(ignore_omp_env: bool = False) -> int:
    """Threads the multi-threaded forms use by default: the OpenMP maximum capped by the CPUs this
    process may run on (cgroup-limited boxes report all host cores to OpenMP).  `ignore_omp_env`: do not cap by
    OMP_NUM_THREADS (torch.distributed.run sets it to 1 for every rank; the `_mt` entry points take their thread
    count as an argument, so the environment does not bind them)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    try:  # cgroup v2 quota, if any
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            avail = min(avail, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    # a one-GPU share of a GPU host is 16 CPUs (the pool's guidance); SKM_HOST_THREADS overrides the cap
    cap = int(os.environ.get("SKM_HOST_THREADS", "16"))
    if ignore_omp_env:
        return max(1, min(avail, cap))
    return max(1, min(avail, cap, int(lib().orc_max_threads())))


def count_csr(rank, nsym, k, seq, off, threads=1):
    """threads=1: the single-threaded restatement; threads=N or 0 (= host_threads()): the OpenMP form
    (row-parallel, same per-row code, identical output)."""
    n = len(off) - 1
    cap = len(seq) + 1
    rowptr = np.zeros(n + 1, dtype=np.int64)
    codes = np.zeros(cap, dtype=np.uint64)
    counts = np.zeros(cap, dtype=np.uint32)
    first = np.zeros(cap, dtype=np.uint32)
    if threads == 1:
        nnz = lib().orc_count_csr(
            _p(rank), nsym, k, _p(seq), _p(off), C.c_int64(n), _p(rowptr), _p(codes), _p(counts), _p(first)
        )
    else:
        nnz = lib().orc_count_csr_mt(
            _p(rank), nsym, k, _p(seq), _p(off), C.c_int64(n), _p(rowptr), _p(codes), _p(counts), _p(first),
            C.c_int(threads or host_threads()),
        )
    return rowptr, codes[:nnz].copy(), counts[:nnz].copy(), first[:nnz].copy()


def basis(rowptr, codes, counts, first, threads=1):
    n, nnz = len(rowptr) - 1, len(codes)
    b = np.zeros(nnz + 1, dtype=np.uint64)
    df = np.zeros(nnz + 1, dtype=np.uint32)
    tot = np.zeros(nnz + 1, dtype=np.uint64)
    fk = np.zeros(nnz + 1, dtype=np.uint64)
    col = np.zeros(nnz + 1, dtype=np.uint32)
    if threads == 1:
        B = lib().orc_basis(
            _p(codes), _p(counts), _p(first), _p(rowptr), C.c_int64(n), C.c_int64(nnz),
            _p(b), _p(df), _p(tot), _p(fk), _p(col),
        )
    else:
        B = lib().orc_basis_mt(
            _p(codes), _p(counts), _p(first), _p(rowptr), C.c_int64(n), C.c_int64(nnz),
            _p(b), _p(df), _p(tot), _p(fk), _p(col), C.c_int(threads or host_threads()),
        )
    return b[:B].copy(), df[:B].copy(), tot[:B].copy(), fk[:B].copy(), col[:nnz].copy()


def cosine_all(rowptr, col, val, ncols, threads=0, keep=False, stats=False):
    """N x N float32 cosine of a CSR count matrix with itself on `threads` threads (0 = host_threads()).
    keep=False: every row is produced in a per-thread buffer and only the sum of all entries is returned
    (the timed cpu_baseline form); keep=True also returns the matrix (tests, small n); stats=True
    returns (total, row sums float64[n], row non-zero counts uint32[n])."""
    n = len(rowptr) - 1
    out = np.zeros((n, n), dtype=np.float32) if keep else None
    rowsum = np.zeros(n, dtype=np.float64) if stats else None
    rownnz = np.zeros(n, dtype=np.uint32) if stats else None
    total = lib().orc_cosine_all_mt(
        C.c_int64(n), _p(np.ascontiguousarray(rowptr, dtype=np.int64)), _p(np.ascontiguousarray(col, dtype=np.uint32)),
        _p(np.ascontiguousarray(val, dtype=np.uint32)), C.c_int64(ncols), _p(out), _p(rowsum), _p(rownnz),
        C.c_int(threads or host_threads()),
    )
    if stats:
        return float(total), rowsum, rownnz
    return (float(total), out) if keep else float(total)


def sampled_gram(rowptr, codes, counts, sample_rows, threads=0):
    """int32 [len(sample_rows), n]: exact dot products of the sampled rows with every row, joined on the
    k-mer codes themselves (no basis needed)."""
    n = len(rowptr) - 1
    sample = np.ascontiguousarray(sample_rows, dtype=np.int64)
    out = np.zeros((len(sample), n), dtype=np.int32)
    lib().orc_sampled_gram_mt(
        C.c_int64(n), _p(np.ascontiguousarray(rowptr, dtype=np.int64)), _p(np.ascontiguousarray(codes, dtype=np.uint64)),
        _p(np.ascontiguousarray(counts, dtype=np.uint32)), _p(sample), C.c_int64(len(sample)), _p(out),
        C.c_int(threads or host_threads()),
    )
    return out


def cosine_rows(xrowptr, xcol, xval, ncols, rows, yrowptr=None, ycol=None, yval=None):
    if yrowptr is None:
        yrowptr, ycol, yval = xrowptr, xcol, xval
    n, m = len(xrowptr) - 1, len(yrowptr) - 1
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    out = np.zeros((len(rows), m), dtype=np.float64)
    xcol = np.ascontiguousarray(xcol, dtype=np.uint32)
    xval = np.ascontiguousarray(xval, dtype=np.uint32)
    ycol = np.ascontiguousarray(ycol, dtype=np.uint32)
    yval = np.ascontiguousarray(yval, dtype=np.uint32)
    xrowptr = np.ascontiguousarray(xrowptr, dtype=np.int64)
    yrowptr = np.ascontiguousarray(yrowptr, dtype=np.int64)
    lib().orc_cosine_rows(
        C.c_int64(n), _p(xrowptr), _p(xcol), _p(xval), C.c_int64(m), _p(yrowptr), _p(ycol), _p(yval),
        C.c_int64(ncols), _p(rows), C.c_int64(len(rows)), _p(out),
    )
    return out
