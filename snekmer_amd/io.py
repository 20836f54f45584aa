"""io: the on-disk formats either side of the hot path.

``load_npz`` / ``read_kmers`` keep the reference contracts (snekmer/io.py:46-96, :99-122);
``save_npz_sparse`` / ``load_counts_npz`` are the sparse variant of the same file (SURVEY.md 8(f)
rank 4): the integer count matrix in CSR form instead of the dense float64 presence matrix
(the reference's memory wall, rules/kmerize.smk:112,132-139); ``load_npz`` reads both;
``read_fasta`` is the minimal reader the rule needs in place of Bio.SeqIO
(rules/kmerize.smk:90-129 only uses ``record.id`` and ``record.seq``).
"""
from os.path import basename, splitext
from typing import Dict, List, Tuple

import numpy as np


def read_fasta(path: str) -> List[Tuple[str, str]]:
    """(id, sequence) per record: id = header up to the first whitespace."""
    records: List[Tuple[str, str]] = []
    name, chunks = None, []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    records.append((name, "".join(chunks)))
                header = line[1:].split()
                name = header[0] if header else ""
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        records.append((name, "".join(chunks)))
    return records


def read_kmers(filename: str) -> List[str]:
    """One k-mer per line, verbatim order (snekmer/io.py:99-122)."""
    with open(filename) as f:
        return [line.strip() for line in f]


def load_npz(
    filename: str,
    columns: Dict[str, str] = {"ids": "sequence_id", "seqs": "sequence", "vecs": "sequence_vector"},
    objects: Tuple = ("kmerlist",),
):
    """``.npz`` written by the vectorize rule -> ([kmerlist], DataFrame) (snekmer/io.py:46-96)."""
    import pandas as pd

    data = np.load(filename)
    if "vecs" not in data.files and "counts_rowptr" in data.files:
        # sparse variant: rebuild the 0/1 presence rows the reference's readers expect
        data = dict(data)
        data["vecs"] = (load_counts_npz(filename).toarray() > 0).astype(np.float64)
    df = {"filename": splitext(basename(filename))[0]}
    for in_col, out_col in columns.items():
        df.update({out_col: list(data[in_col])})
        if "seq" in in_col:
            df.update({f"{out_col}_length": [len(s) for s in data[in_col]]})
    extras = [data[obj] for obj in objects]
    return extras, pd.DataFrame(df)


SPARSE_KEYS = ("kmerlist", "ids", "seqs", "lengths", "counts_rowptr", "counts_col", "counts_val")


def save_npz_sparse(filename: str, out: Dict[str, np.ndarray]) -> None:
    """The rule's ``.npz`` with the count matrix as CSR (`counts_rowptr/col/val` over the columns
    of ``kmerlist``) in place of the dense presence matrix ``vecs``.  `out` is what
    kmerize.vectorize_records returns."""
    np.savez_compressed(filename, **{k: out[k] for k in SPARSE_KEYS})


def load_counts_npz(filename: str):
    """scipy.sparse.csr_matrix of integer k-mer counts [sequences x kmerlist] from a sparse ``.npz``
    (the matrix rules/learn.smk:359-383 and rules/apply.smk:188-206 rebuild in Python)."""
    from scipy.sparse import csr_matrix

    data = np.load(filename)
    n, b = len(data["ids"]), len(data["kmerlist"])
    return csr_matrix((data["counts_val"].astype(np.int64), data["counts_col"].astype(np.int64), data["counts_rowptr"]),
                      shape=(n, b))
