"""Shared test helpers: golden loading and packing (tests only)."""
import json
import os

import numpy as np

from snekmer_amd import alphabet as skm_alphabet
from snekmer_amd.utils import pack_sequences

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ALL_ALPHABETS = ["hydro", "standard", "solvacc", "hydrocharge", "hydrostruct", "miqs", "ptm", "None", "red6"]


def ensure_red6():
    if "red6" not in skm_alphabet.ALPHABETS:
        skm_alphabet.register_alphabet("red6", skm_alphabet.RED6_GROUPS)


def gjson(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def gnpz(name):
    return np.load(os.path.join(GOLDEN, name))


def demo_records():
    from oracle.ref_path import read_fasta

    recs = []
    for f in ("TIGR03149.faa", "nxrA.faa"):
        recs += read_fasta(os.path.join(GOLDEN, "data", f))
    return recs


def csr_to_dense(rowptr, col, val, ncols):
    n = len(rowptr) - 1
    M = np.zeros((n, ncols), dtype=np.int64)
    for i in range(n):
        s, e = int(rowptr[i]), int(rowptr[i + 1])
        M[i, np.asarray(col[s:e], dtype=np.int64)] = val[s:e]
    return M


def alpha_key(name):
    return None if name == "None" else name


def parse_tag(tag):
    a, k, mf = tag.rsplit("_", 2)
    return alpha_key(a), int(k[1:]), int(mf[2:])
