"""alphabet: amino-acid recoding (AAR) tables and the byte LUTs the HIP kernels consume.

Mirrors the public surface of the reference module ``snekmer/alphabet.py``
(tables :13-105, ``check_valid`` :121-155, ``get_alphabet`` :158-197,
``get_alphabet_name`` :200-238, ``get_alphabet_keys`` :241-266) so callers such as
``rules/kmerize.smk:67,122-126`` work unchanged.  On top of that surface this module
adds what the device path needs: :func:`build_lut` turns an alphabet into two
256-entry byte tables (translate table + class-rank table) and
:func:`register_alphabet` lets benchmarks add a non-reference alphabet (``red6``).

The tables themselves are data and must equal the reference's; they are written here
class-first (class letter <- residues, in the reference's group order) and expanded
into the reference's dict shapes at import time.
"""
from typing import Dict, Iterable, List, Sequence, Set, Tuple, Union

import numpy as np

StandardAlphabet = "AILMVFYWSTQNCHDEKRGP"
AA_SELF_MAPPING = {a: a for a in StandardAlphabet}

PTM_CHARS = "-_!^#$@.%&"
PTM_SELF_MAPPING = {c: c for c in PTM_CHARS}

ALPHABET_ORDER = {
    0: "hydro",
    1: "standard",
    2: "solvacc",
    3: "hydrocharge",
    4: "hydrostruct",
    5: "miqs",
}

# (class letter, residues) in the reference's group order; the trailing string is the
# reference's "_keys" entry (an authorial letter order that no code path consumes).
_GROUPED = {
    "hydro": ([("S", "SFTNKYEQCWPHDR"), ("V", "VMLAIG")], "SV"),
    "standard": (
        [
            ("A", "AGILMV"),
            ("P", "PH"),
            ("F", "FWY"),
            ("N", "NQST"),
            ("D", "DE"),
            ("K", "KR"),
            ("C", "C"),
        ],
        "APFNDKC",
    ),
    "solvacc": ([("C", "CILMVFWY"), ("A", "AGHST"), ("P", "PDEKNQR")], "CAP"),
    # NB: no group holds "E", and "N" sits in two groups (the later one wins).
    "hydrocharge": ([("L", "SFTNYQCWPH"), ("H", "VMLAIG"), ("C", "KNDR")], "LHC"),
    "hydrostruct": ([("L", "SFTNKYEQCWHDR"), ("H", "VMLAI"), ("B", "PG")], "LHB"),
    "miqs": (
        [
            ("A", "A"),
            ("C", "C"),
            ("D", "DEN"),
            ("F", "FWY"),
            ("G", "G"),
            ("H", "H"),
            ("I", "ILMQV"),
            ("K", "KR"),
            ("P", "P"),
            ("S", "ST"),
        ],
        "ACDFGHIKPS",
    ),
}


def _as_reference_dict(groups: Sequence[Tuple[str, str]], keys: str) -> Dict[str, str]:
    d = {residues: letter for letter, residues in groups}
    d["_keys"] = keys
    return d


ALPHABETS: Dict[str, Dict[str, str]] = {
    name: _as_reference_dict(groups, keys) for name, (groups, keys) in _GROUPED.items()
}
ALPHABETS["ptm"] = {
    **AA_SELF_MAPPING,
    **PTM_SELF_MAPPING,
    "_keys": StandardAlphabet + PTM_CHARS,
}
ALPHABETS["None"] = AA_SELF_MAPPING


def _expand(mapping: Dict[str, str]) -> Dict[str, str]:
    """Group-keyed dict -> per-residue dict; later groups override earlier ones."""
    out: Dict[str, str] = {}
    for group, letter in mapping.items():
        if group == "_keys":
            continue
        for residue in group:
            out[residue] = letter
    return out


FULL_ALPHABETS: Dict[str, Dict[str, str]] = {a: _expand(m) for a, m in ALPHABETS.items()}

ALPHABET_ID = {
    f"RED{n}": {v: k for k, v in ALPHABETS[ALPHABET_ORDER[n]].items()}
    for n in range(len(ALPHABET_ORDER))
}
ALPHABET2ID = {ALPHABET_ORDER[n]: f"RED{n}" for n in range(len(ALPHABET_ORDER))}


def get_alphabets() -> Dict[str, Dict[str, str]]:
    return ALPHABETS


def check_valid(alphabet: Union[str, int]) -> None:
    """Raise ValueError unless `alphabet` is a known name, a small int, or None.

    Same acceptance test as the reference (alphabet.py:136-154), including its quirk
    that any int below ``len(ALPHABETS)`` passes here and fails later with KeyError.
    """
    known = (alphabet in range(len(ALPHABETS))) or (alphabet in ALPHABETS)
    if not known and str(alphabet) != "None":
        raise ValueError(
            "Invalid alphabet specified; alphabet must be a"
            " string (see snekmer.alphabet) or integer"
            " n between"
            f" {min(list(ALPHABET_ORDER.keys()))}"
            " and"
            f" {max(list(ALPHABET_ORDER.keys()))}"
            "."
        )


def _resolve_name(alphabet: Union[str, int, None]) -> str:
    check_valid(alphabet)
    if alphabet is None:
        alphabet = str(alphabet)
    if isinstance(alphabet, int):
        alphabet = ALPHABET_ORDER[alphabet]  # KeyError for 6, 7 as in the reference
    return alphabet


def get_alphabet(alphabet: Union[str, int], mapping: dict = ALPHABETS) -> Dict[str, str]:
    return mapping[_resolve_name(alphabet)]


def get_alphabet_name(alphabet: Union[str, int], mapping: dict = ALPHABETS) -> str:
    return _resolve_name(alphabet)


def get_alphabet_keys(
    alphabet: Union[str, int], mapping: Dict[str, dict] = FULL_ALPHABETS
) -> Set[str]:
    if alphabet is None:
        alphabet = str(alphabet)
    alphabet_map = get_alphabet(alphabet, mapping)
    if "_keys" in alphabet_map.keys():
        alphabet_map.pop("_keys")
    return set(alphabet_map.values())


# --------------------------------------------------------------------------------------
# Additions for the device path
# --------------------------------------------------------------------------------------
RED6_GROUPS = (
    ("A", "AGILMV"),
    ("P", "PH"),
    ("F", "FWY"),
    ("N", "NQSTC"),
    ("D", "DE"),
    ("K", "KR"),
)
"""Benchmark alphabet `red6` (SURVEY.md 8(d)): the reference's `standard` with the
singleton class C folded into polar N.  NOT a reference feature (SURVEY.md D1)."""


def register_alphabet(name: str, groups: Iterable[Tuple[str, str]], keys: str = None) -> None:
    """Add a user alphabet to ALPHABETS / FULL_ALPHABETS (both plain dicts, as upstream)."""
    groups = list(groups)
    ALPHABETS[name] = _as_reference_dict(groups, keys or "".join(g[0] for g in groups))
    FULL_ALPHABETS[name] = _expand(ALPHABETS[name])


INVALID = 0xFF


class AlphabetLUT:
    """Byte tables for one alphabet.

    translate[b] : byte after recoding (unmapped bytes pass through, vectorize.py:195)
    rank[b]      : rank of translate[b] among the class letters in ASCII order, or 0xFF
                   when translate[b] is not a class letter (window invalid, vectorize.py:247)
    letters      : class letters in rank order; code = sum rank_i * nsym**(k-1-i), so integer
                   order == lexicographic k-mer string order.
    """

    def __init__(self, translate: np.ndarray, rank: np.ndarray, letters: str):
        self.translate = translate
        self.rank = rank
        self.letters = letters
        self.nsym = len(letters)

    def code_bits(self, k: int) -> int:
        """32 or 64: narrowest code word with the all-ones pattern left free as sentinel."""
        space = self.nsym**k
        if space < 2**32:
            return 32
        if space < 2**64:
            return 64
        raise ValueError(
            f"k-mer space {self.nsym}^{k} does not fit a 64-bit code word; unsupported"
        )

    def decode(self, codes: np.ndarray, k: int) -> np.ndarray:
        """Integer codes -> numpy '<U{k}' k-mer strings (host-side formatting only)."""
        codes = np.asarray(codes, dtype=np.uint64)
        if codes.size == 0:
            return np.array([], dtype=str)
        # G symbols per step through a table of all G-letter words (G chosen so the table stays small)
        G = 1
        while G < 4 and self.nsym ** (G + 1) <= 1 << 16:
            G += 1
        table = self.__dict__.get("_words")
        if table is None or table.shape[1] != G:
            letters = np.frombuffer(self.letters.encode("latin-1"), dtype=np.uint8).astype(np.uint32)
            idx = np.arange(self.nsym**G)
            table = np.empty((idx.size, G), dtype=np.uint32)  # UCS-4 code points: the result is VIEWED as unicode
            for g in range(G - 1, -1, -1):
                table[:, g] = letters[idx % self.nsym]
                idx = idx // self.nsym
            self.__dict__["_words"] = table
        out = np.empty((codes.size, k), dtype=np.uint32)
        rem = codes.copy()
        step = np.uint64(self.nsym**G)
        i = k
        while i >= G:
            out[:, i - G : i] = table[(rem % step).astype(np.intp)]
            rem //= step
            i -= G
        if i:  # the leading k mod G symbols
            out[:, :i] = table[rem.astype(np.intp)][:, G - i :]
        return out.view(f"<U{k}").ravel()

    def encode(self, kmers: Sequence[str], k: int) -> Tuple[np.ndarray, np.ndarray]:
        """k-mer strings -> (codes uint64, ok mask). ok is False for strings that are not
        k class letters long (such strings can never match a window)."""
        codes = np.zeros(len(kmers), dtype=np.uint64)
        ok = np.ones(len(kmers), dtype=bool)
        pos = {c: i for i, c in enumerate(self.letters)}
        for idx, s in enumerate(kmers):
            s = str(s)
            if len(s) != k or any(ch not in pos for ch in s):
                ok[idx] = False
                continue
            c = 0
            for ch in s:
                c = c * self.nsym + pos[ch]
            codes[idx] = c
        return codes, ok


def build_lut(alphabet: Union[str, int, None], mapping: Dict[str, dict] = FULL_ALPHABETS) -> AlphabetLUT:
    """Build the byte LUTs for `alphabet`.

    Only 1-byte -> 1-byte maps are representable; the reference's ``str.translate`` would
    also accept multi-character values in a user-supplied `mapping`, which is rejected here.
    """
    table = dict(get_alphabet(alphabet, mapping))
    table.pop("_keys", None)
    char_set = set(table.values())
    for src, dst in table.items():
        if len(src) != 1 or len(dst) != 1 or ord(src) > 255 or ord(dst) > 255:
            raise ValueError(
                "alphabet maps must be single latin-1 characters on both sides "
                f"(got {src!r}->{dst!r})"
            )
    if len(char_set) >= INVALID:
        raise ValueError("alphabets with more than 254 classes are unsupported")
    letters = "".join(sorted(char_set))
    translate = np.arange(256, dtype=np.uint8)
    for src, dst in table.items():
        translate[ord(src)] = ord(dst)
    rank = np.full(256, INVALID, dtype=np.uint8)
    for b in range(256):
        t = chr(int(translate[b]))
        if t in char_set:
            rank[b] = letters.index(t)
    return AlphabetLUT(translate, rank, letters)
