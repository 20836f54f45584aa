"""io: the on-disk formats either side of the hot path.

``load_npz`` / ``read_kmers`` keep the reference contracts (snekmer/io.py:46-96, :99-122);
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
    df = {"filename": splitext(basename(filename))[0]}
    for in_col, out_col in columns.items():
        df.update({out_col: list(data[in_col])})
        if "seq" in in_col:
            df.update({f"{out_col}_length": [len(s) for s in data[in_col]]})
    extras = [data[obj] for obj in objects]
    return extras, pd.DataFrame(df)
