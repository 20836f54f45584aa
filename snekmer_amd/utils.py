"""utils: host-side helpers on the hot path's boundary.

``check_list`` and ``to_feature_matrix`` keep the reference's contracts
(snekmer/utils.py:77-93 and :183-203).  ``pack_sequences`` builds the packed
``bytes + offsets[N+1]`` layout every device entry point consumes.
"""
import collections.abc
from typing import Any, Iterable, List, Optional, Tuple, Union

import numpy as np

try:  # pandas is optional for the hot path; the reference accepts Series where present
    import pandas as pd

    _SERIES = (pd.Series,)
except Exception:  # pragma: no cover
    _SERIES = ()

SUBSTITUTE = 0x1A  # stands in for characters outside latin-1; never a class letter


def check_list(array: Any) -> bool:
    """True for sequences, ndarrays and Series (snekmer/utils.py:77-93)."""
    return isinstance(array, (collections.abc.Sequence, np.ndarray) + _SERIES)


def to_feature_matrix(array, length_array=None) -> np.ndarray:
    """Rows -> 2-D array, each row divided by its entry of `length_array` (default 1)
    (snekmer/utils.py:183-203)."""
    if length_array is None:
        length_array = np.ones(len(array))
    return np.asarray([np.array(a) / length for a, length in zip(array, length_array)])


def _encode(seq: str) -> bytes:
    try:
        return seq.encode("latin-1")
    except UnicodeEncodeError:
        return bytes(ord(c) if ord(c) < 256 else SUBSTITUTE for c in seq)


def pack_sequences(seqs: Iterable[Union[str, bytes]]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate sequences into one uint8 buffer plus int64 offsets[N+1]."""
    chunks: List[bytes] = []
    for s in seqs:
        chunks.append(bytes(s) if isinstance(s, (bytes, bytearray)) else _encode(str(s)))
    offsets = np.zeros(len(chunks) + 1, dtype=np.int64)
    if chunks:
        np.cumsum([len(c) for c in chunks], out=offsets[1:])
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
    return data, offsets


def unpack_sequences(data: np.ndarray, offsets: np.ndarray, lengths: Optional[np.ndarray] = None) -> List[str]:
    """Inverse of pack_sequences; `lengths` (if given) truncates each record."""
    raw = np.ascontiguousarray(data, dtype=np.uint8).tobytes()
    out = []
    for i in range(len(offsets) - 1):
        b = int(offsets[i])
        e = b + int(lengths[i]) if lengths is not None else int(offsets[i + 1])
        out.append(raw[b:e].decode("latin-1"))
    return out
