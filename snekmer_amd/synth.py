"""synth: deterministic synthetic protein families for benchmarks and parity tests.

The reference ships no benchmark data.  Independent random
sequences share no 12-mers, which would make the pairwise step vacuous, so the generator
emits *families*: `n // family` random roots, each member a copy of its root with
per-residue substitutions.  1 % of members carry one ``X`` (exercises
invalid-window skipping, snekmer/vectorize.py:247) and 1 % a trailing ``*`` (exercises the
``rstrip("*")`` of snekmer/vectorize.py:193).  Members are emitted in a seeded random
order, not family by family, as a real FASTA would be.

Everything is a function of (n, length, family, p_sub, seed) alone.
"""
from typing import List, Tuple

import numpy as np

RESIDUES = "ARNDCQEGHILKMFPSTWYV"
# Approximate UniProtKB/Swiss-Prot amino-acid composition (percent), same order as RESIDUES.
BACKGROUND_PERCENT = (
    8.25, 5.53, 4.06, 5.45, 1.37, 3.93, 6.75, 7.07, 2.27, 5.96,
    9.66, 5.84, 2.42, 3.86, 4.70, 6.56, 5.34, 1.08, 2.92, 6.87,
)
BASE_SEED = 20250523
_CHUNK = 20000


def synth_families(
    n: int,
    length: int = 300,
    family: int = 100,
    p_sub: float = 0.10,
    seed: int = BASE_SEED,
    shuffle: bool = True,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Return (residues uint8[total], offsets int64[n+1], family_id int32[n])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bg = np.asarray(BACKGROUND_PERCENT, dtype=np.float64)
    bg = bg / bg.sum()
    alphabet = np.frombuffer(RESIDUES.encode(), dtype=np.uint8)

    n_fam = max(1, -(-n // family))
    roots = rng.choice(20, size=(n_fam, length), p=bg).astype(np.uint8)
    fam_of = (np.arange(n, dtype=np.int64) // family).astype(np.int32)
    if shuffle:
        fam_of = fam_of[rng.permutation(n)]

    has_x = rng.random(n) < 0.01
    x_pos = rng.integers(0, length, size=n)
    has_star = rng.random(n) < 0.01

    lens = np.full(n, length, dtype=np.int64) + has_star
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    out = np.empty(int(offsets[-1]), dtype=np.uint8)

    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        m = hi - lo
        body = roots[fam_of[lo:hi]]
        mask = rng.random((m, length)) < p_sub
        repl = rng.choice(20, size=(m, length), p=bg).astype(np.uint8)
        body = np.where(mask, repl, body)
        chars = alphabet[body]
        rows = np.nonzero(has_x[lo:hi])[0]
        chars[rows, x_pos[lo:hi][rows]] = ord("X")
        # scatter rows into the ragged output
        starts = offsets[lo:hi]
        idx = starts[:, None] + np.arange(length)[None, :]
        out[idx.ravel()] = chars.ravel()
        star_rows = np.nonzero(has_star[lo:hi])[0]
        out[starts[star_rows] + length] = ord("*")
    return out, offsets, fam_of


def synth_skewed(
    n: int,
    seed: int = BASE_SEED,
    max_family: int = 5000,
    len_median: float = 300.0,
    len_sigma: float = 0.6,
    len_min: int = 50,
    len_max: int = 5000,
    p_sub: float = 0.08,
    p_indel: float = 0.02,
    p_lowcomplexity: float = 0.05,
    zipf_a: float = 1.6,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A batch with the skew real proteomes have and `synth_families` lacks: family sizes Zipf-distributed between 1
    and `max_family`, root lengths log-normal clipped to [len_min, len_max], members = root with substitutions AND
    indels (each position deleted or followed by an inserted residue with probability p_indel / 2 each), 5 % of the
    sequences carrying a low-complexity insert (a homopolymer or dipeptide run of 10-60 residues), 1 % an 'X', 1 % a
    trailing '*'; members in a seeded random order.  Returns (residues uint8[total], offsets int64[n+1], family
    int32[n]); a function of the arguments alone."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bg = np.asarray(BACKGROUND_PERCENT, dtype=np.float64)
    bg = bg / bg.sum()
    alphabet = np.frombuffer(RESIDUES.encode(), dtype=np.uint8)
    sizes = []
    left = n
    while left > 0:
        s = int(min(rng.zipf(zipf_a), max_family, left))
        sizes.append(s)
        left -= s
    fam_of = np.repeat(np.arange(len(sizes), dtype=np.int32), sizes)
    fam_of = fam_of[rng.permutation(n)]
    root_len = np.clip(np.round(np.exp(rng.normal(np.log(len_median), len_sigma, size=len(sizes)))), len_min, len_max).astype(np.int64)
    roots = [alphabet[rng.choice(20, size=int(L), p=bg)] for L in root_len]
    seqs = []
    for i in range(n):
        body = roots[fam_of[i]].copy()
        L = body.size
        sub = rng.random(L) < p_sub
        body[sub] = alphabet[rng.choice(20, size=int(sub.sum()), p=bg)]
        u = rng.random(L)
        keep = u >= p_indel / 2
        ins = (u >= p_indel / 2) & (u < p_indel)
        if ins.any() or not keep.all():
            # every kept position is followed by one inserted residue where `ins` is set
            reps = keep.astype(np.int64) + ins.astype(np.int64)
            out = np.repeat(body, reps)
            at = np.cumsum(reps)[ins] - 1
            out[at] = alphabet[rng.choice(20, size=at.size, p=bg)]
            body = out
        if rng.random() < p_lowcomplexity and body.size:
            run = int(rng.integers(10, 61))
            unit = alphabet[rng.choice(20, size=int(rng.integers(1, 3)), p=bg)]
            pos = int(rng.integers(0, body.size + 1))
            body = np.concatenate([body[:pos], np.tile(unit, run)[:run], body[pos:]])
        if rng.random() < 0.01 and body.size:
            body[int(rng.integers(0, body.size))] = ord("X")
        if rng.random() < 0.01:
            body = np.concatenate([body, np.frombuffer(b"*", dtype=np.uint8)])
        seqs.append(body)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([s.size for s in seqs], out=offsets[1:])
    return (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)).astype(np.uint8), offsets, fam_of


def to_records(residues: np.ndarray, offsets: np.ndarray, prefix: str = "syn") -> List[Tuple[str, str]]:
    """(id, sequence) pairs, the shape a FASTA reader yields."""
    raw = residues.tobytes()
    return [
        (f"{prefix}{i:07d}", raw[int(offsets[i]) : int(offsets[i + 1])].decode("latin-1"))
        for i in range(len(offsets) - 1)
    ]


def write_fasta(path: str, records: List[Tuple[str, str]], width: int = 60) -> None:
    with open(path, "w") as fh:
        for rid, seq in records:
            fh.write(f">{rid}\n")
            for i in range(0, len(seq), width):
                fh.write(seq[i : i + width] + "\n")
