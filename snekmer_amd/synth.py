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
