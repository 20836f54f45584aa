"""ORACLE (test infrastructure, not product code).

CPU restatement, in plain Python/numpy, of the reference's AAR-kmer vectorize + count +
cosine path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file, and only as the checker / reported baseline;
nothing under ``snekmer_amd/`` imports it.

Parity status: PINNED.  Every function here is checked against outputs of the imported
reference (``tests/golden/make_golden.py`` run in the build container, fixtures committed
under ``tests/golden/``) by ``tests/test_oracle_golden.py``.

Each function cites the reference lines it follows (paths relative to the upstream repo).
The algorithmic shape is kept deliberately: per-sequence ``np.isin`` over the string basis
and the O(N*|basis|) count projection are what the reference does, so timing this file is
timing "Snekmer's own numpy path" (baseline 1).
"""
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np


# ----------------------------------------------------------------------------- FASTA
def read_fasta(path: str) -> List[Tuple[str, str]]:
    """Minimal stand-in for ``Bio.SeqIO.parse(path, "fasta")`` as used by
    rules/kmerize.smk:90-129: id = header up to first whitespace, seq = joined lines."""
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


# ----------------------------------------------------------------------------- recode
def reduce(sequence: str, table: Dict[str, str]) -> str:
    """snekmer/vectorize.py:193-195 — strip trailing '*', then str.translate; characters
    without a map entry pass through unchanged."""
    sequence = str(sequence).rstrip("*")
    return sequence.translate(sequence.maketrans(table))


def kmer_gen(reduced: str, k: int, char_set: set) -> Iterable[str]:
    """snekmer/vectorize.py:239-249 — every window whose characters are all class letters."""
    i = 0
    n = len(reduced) - k + 1
    while i < n:
        kmer = reduced[i : i + k]
        if set(kmer) <= char_set:
            yield kmer
        i += 1


def reduce_vectorize(sequence: str, k: int, table: Dict[str, str]) -> np.ndarray:
    """snekmer/vectorize.py:292-328 — k-mer strings, window order, duplicates kept."""
    char_set = set(table.values())
    return np.array(list(kmer_gen(reduce(sequence, table), k, char_set)), dtype=str)


# ----------------------------------------------------------------------------- rule body
def kmerize_rule(
    records: Sequence[Tuple[str, str]],
    k: int,
    table: Dict[str, str],
    min_filter: int = 0,
    basis: Optional[Sequence[str]] = None,
) -> Dict[str, np.ndarray]:
    """rules/kmerize.smk:67-139 — observed basis in first-seen order (kept iff total
    occurrences > min_filter), binary presence matrix via np.isin, reduced strings, ids,
    raw lengths.  With `basis` given (the rule's input/basis.txt branch, :72-78) the
    basis is taken verbatim and min_filter is ignored."""
    if basis is not None:
        kmerbasis = list(basis)
    else:
        seen: Dict[str, int] = {}
        for _, seq in records:
            for key in reduce_vectorize(seq, k, table):
                if key in seen:
                    seen[key] += 1
                else:
                    seen[key] = 1
        kmerbasis = np.array(list(seen.keys()))[np.array(list(seen.values())) > min_filter]

    vecs = np.zeros((len(records), len(kmerbasis)))
    seqs, ids, lengths = [], [], []
    for n, (rid, seq) in enumerate(records):
        addvec = reduce_vectorize(seq, k, table)
        vecs[n][np.isin(kmerbasis, addvec)] = 1
        seqs.append(reduce(seq, table))
        ids.append(rid)
        lengths.append(len(seq))
    return {
        "kmerlist": np.asarray(kmerbasis),
        "ids": np.asarray(ids),
        "seqs": np.asarray(seqs),
        "vecs": vecs,
        "lengths": np.asarray(lengths),
    }


def count_matrix(reduced_seqs: Sequence[str], kmerlist: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """rules/learn.smk:359-383 and rules/apply.smk:188-206 — for each *reduced* string count
    every length-k substring (no validity test), project onto `kmerlist` order, and keep
    running column totals."""
    kmerlist = [str(x) for x in kmerlist]
    k_len = len(kmerlist[0]) if len(kmerlist) else 0
    totals = [0] * len(kmerlist)
    rows = []
    for v in reduced_seqs:
        v = str(v)
        k_counts: Dict[str, int] = {}
        for item in range(0, len(v) - k_len + 1):
            j = v[item : item + k_len]
            k_counts[j] = k_counts.get(j, 0) + 1
        store = [k_counts.get(item, 0) for item in kmerlist]
        for i, item in enumerate(kmerlist):
            totals[i] += k_counts.get(item, 0)
        rows.append(store)
    counts = np.asarray(rows, dtype=np.int64).reshape(len(rows), len(kmerlist))
    return counts, np.asarray(totals, dtype=np.int64)


# ----------------------------------------------------------------------------- cosine
def cosine_similarity(X, Y=None) -> np.ndarray:
    """scikit-learn ``metrics.pairwise.cosine_similarity`` (the reference's un-vendored,
    unpinned dependency: requirements.txt:8; call sites rules/apply.smk:282-284,
    rules/learn.smk:821-823, rules/evaluate.smk:434-436).  Published algorithm (sklearn
    1.7.2 inspected): cast to float64, L2-normalise rows with zero norms
    replaced by 1, then X_hat @ Y_hat.T."""
    X = np.asarray(X, dtype=np.float64)
    Y = X if Y is None else np.asarray(Y, dtype=np.float64)

    def _normalize(A):
        norms = np.sqrt(np.einsum("ij,ij->i", A, A))
        norms[norms == 0.0] = 1.0
        return A / norms[:, np.newaxis]

    Xn = _normalize(X)
    Yn = Xn if Y is X else _normalize(Y)
    return Xn @ Yn.T


def cosine_distances(X) -> np.ndarray:
    """sklearn ``pairwise_distances(X, metric="cosine")`` as reached from
    snekmer/score.py:169-171: 1 - cos, clipped to [0, 2], exact-zero diagonal."""
    S = cosine_similarity(X)
    S *= -1
    S += 1
    np.clip(S, 0, 2, out=S)
    np.fill_diagonal(S, 0.0)
    return S


def hamming_similarity(X) -> np.ndarray:
    """snekmer/score.py:166-168 — the reference's metric="jaccard" branch really computes
    1 - hamming distance (fraction of differing columns)."""
    X = np.asarray(X)
    n, m = X.shape
    out = np.empty((n, n), dtype=np.float64)
    for i in range(n):
        out[i] = 1.0 - (X != X[i]).sum(axis=1) / float(m)
    return out


def jaccard_distance(X) -> np.ndarray:
    """scipy's pdist(X, "jaccard") + squareform as snekmer/scripts/cluster_cluster.py:189-190 calls it.  scipy
    (1.15, the installed un-pinned dependency) reads numeric rows as booleans (non-zero = True):
    |a xor b| / |a or b|, 0 for two empty rows."""
    nz = np.asarray(X) != 0
    n = nz.shape[0]
    out = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        num = (nz ^ nz[i]).sum(axis=1).astype(np.float64)
        den = (nz | nz[i]).sum(axis=1).astype(np.float64)
        out[i] = np.where(den > 0, num / np.maximum(den, 1.0), 0.0)
    return out


def apply_epilogue(totals, counts, names: Sequence[str], confidence: Optional[Dict[float, float]] = None):
    """rules/apply.smk:278-328 (same logic at rules/learn.smk:811-849): cosine of the family totals
    against the query counts, the two best families per query by ``np.argsort(-S)``, Score = top1,
    delta = round(top1 - top2, 2) and Confidence = the global table's entry for that delta (NaN when
    the table has no such key, as ``Series.map`` gives)."""
    S = cosine_similarity(np.asarray(totals, dtype=np.float64), np.asarray(counts, dtype=np.float64)).T
    sorted_vals = np.argsort(-S, axis=1)[:, :2]
    score_rank = np.take_along_axis(S, sorted_vals, axis=1)
    delta = np.round(score_rank[:, 0] - score_rank[:, 1], 2)
    out = {
        "sorted_vals": sorted_vals,
        "score_rank": score_rank,
        "Score": score_rank[:, 0],
        "delta": delta,
        "Prediction": np.asarray([str(names[i]) for i in sorted_vals[:, 0]]),
    }
    if confidence is not None:
        table = {float(k): float(v) for k, v in confidence.items()}
        out["Confidence"] = np.asarray([table.get(float(d), np.nan) for d in delta], dtype=np.float64)
    return out


# ----------------------------------------------------------------------------- basis ops
def basis_transform(basis: Sequence[str], vector, vector_basis: Sequence[str]) -> np.ndarray:
    """snekmer/vectorize.py:54-119 — re-index the columns of `vector` (given in
    `vector_basis` order) into `basis` order; k-mers missing from `vector_basis` become
    zero columns."""
    vector = np.asarray(vector)
    try:
        vector_size = vector.shape[1]
    except IndexError:
        vector_size = len(vector)
    if vector_size != len(vector_basis):
        raise ValueError("shape mismatch")
    where = {kmer: i for i, kmer in enumerate(vector_basis)}
    padded = np.insert(vector, vector.shape[1], [0] * vector.shape[0], axis=1)
    idx = [where.get(kmer, padded.shape[1] - 1) for kmer in basis]
    return padded[:, idx]


def make_feature_matrix(vecs: Sequence[Sequence[str]], min_filter: int = 1):
    """snekmer/vectorize.py:201-221 — sorted unique k-mers with total count > min_filter,
    binary rows via np.isin."""
    kmerlist: List[str] = []
    for this in vecs:
        kmerlist.extend(this)
    kmerlist, kmercounts = np.unique(kmerlist, return_counts=True)
    kmerlist = kmerlist[kmercounts > min_filter]
    nk = len(kmerlist)
    result = []
    for i in range(len(vecs)):
        this = np.zeros(nk)
        this[np.isin(kmerlist, vecs[i])] = 1
        result.append(this)
    return result, kmerlist


# ----------------------------------------------------------------------------- end to end
def vectorize_and_cosine(records, k: int, table: Dict[str, str]):
    """The reference CPU path the headline metric is quoted against: kmerize rule body,
    count matrix, N x N cosine.  Returns (rule outputs, counts, S)."""
    out = kmerize_rule(records, k, table)
    counts, _ = count_matrix(out["seqs"], out["kmerlist"])
    S = cosine_similarity(counts) if counts.shape[1] else np.zeros((len(records),) * 2)
    return out, counts, S
