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
        _LIB = C.CDLL(build())
        _LIB.orc_count_csr.restype = C.c_int64
        _LIB.orc_basis.restype = C.c_int64
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


def count_csr(rank, nsym, k, seq, off):
    n = len(off) - 1
    cap = len(seq) + 1
    rowptr = np.zeros(n + 1, dtype=np.int64)
    codes = np.zeros(cap, dtype=np.uint64)
    counts = np.zeros(cap, dtype=np.uint32)
    first = np.zeros(cap, dtype=np.uint32)
    nnz = lib().orc_count_csr(
        _p(rank), nsym, k, _p(seq), _p(off), C.c_int64(n), _p(rowptr), _p(codes), _p(counts), _p(first)
    )
    return rowptr, codes[:nnz].copy(), counts[:nnz].copy(), first[:nnz].copy()


def basis(rowptr, codes, counts, first):
    n, nnz = len(rowptr) - 1, len(codes)
    b = np.zeros(nnz + 1, dtype=np.uint64)
    df = np.zeros(nnz + 1, dtype=np.uint32)
    tot = np.zeros(nnz + 1, dtype=np.uint64)
    fk = np.zeros(nnz + 1, dtype=np.uint64)
    col = np.zeros(nnz + 1, dtype=np.uint32)
    B = lib().orc_basis(
        _p(codes), _p(counts), _p(first), _p(rowptr), C.c_int64(n), C.c_int64(nnz),
        _p(b), _p(df), _p(tot), _p(fk), _p(col),
    )
    return b[:B].copy(), df[:B].copy(), tot[:B].copy(), fk[:B].copy(), col[:nnz].copy()


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
