"""io: the on-disk formats either side of the hot path.

``load_npz`` / ``read_kmers`` keep the reference contracts (snekmer/io.py:46-96, :99-122);
``save_npz_sparse`` / ``load_counts_npz`` are the sparse variant of the same file (SURVEY.md 8(f)
rank 4): the integer count matrix in CSR form instead of the dense float64 presence matrix
(the reference's memory wall, rules/kmerize.smk:112,132-139); ``load_npz`` reads both;
``read_fasta`` is the minimal reader the rule needs in place of Bio.SeqIO
(rules/kmerize.smk:90-129 only uses ``record.id`` and ``record.seq``).
"""
import importlib
import pickle
import sys
import threading
import types
from os.path import basename, splitext
from typing import Any, Dict, List, Tuple

import numpy as np


def read_fasta(path: str) -> List[Tuple[str, str]]:
    """(id, sequence) per record, as ``Bio.SeqIO.parse(path, "fasta")`` yields ``record.id`` / ``str(record.seq)``
    (Biopython's SimpleFastaParser: text before the first '>' line is skipped; id = first word of the header; the
    sequence = the record's lines without their trailing whitespace, joined, with every space and '\\r' removed).
    Text-mode reference implementation; `read_fasta_packed` is the fast form."""
    records: List[Tuple[str, str]] = []
    name, chunks = None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    records.append((name, "".join(chunks).replace(" ", "").replace("\r", "")))
                header = line[1:].split(None, 1)
                name = header[0] if header else ""
                chunks = []
            elif name is not None:
                chunks.append(line.rstrip())
    if name is not None:
        records.append((name, "".join(chunks).replace(" ", "").replace("\r", "")))
    return records


def read_fasta_packed(path: str, threads: int = 0, with_records: bool = False):
    """FASTA file -> (ids '<U' array, residues uint8[total], offsets int64[n+1]): the packed layout the device entry
    points take, produced by the C library's threaded reader (skm_fasta_index / skm_fasta_parse: host code, no GPU)
    instead of a per-record Python loop.  Same records as `read_fasta`; files with non-ASCII bytes (which a text
    handle decodes as UTF-8) take the text-mode path.  `with_records=True` adds a fourth item: the (id, sequence)
    strings when the text-mode path was taken (the packed bytes cannot hold characters above latin-1), else None."""
    import ctypes as C

    from . import _hip
    from .utils import pack_sequences

    lib = _hip.load_library()
    with open(path, "rb") as fh:
        buf = np.frombuffer(fh.read(), dtype=np.uint8)
    p = C.c_void_p
    nrec, nres, flags = C.c_int64(0), C.c_int64(0), C.c_int(0)
    _hip._check(lib, lib.skm_fasta_index(buf.ctypes.data_as(p), buf.size, threads, C.byref(nrec), C.byref(nres), C.byref(flags)))
    if flags.value & 1:
        recs = read_fasta(path)
        res, off = pack_sequences([s for _, s in recs])
        ids = np.asarray([r[0] for r in recs], dtype=str) if recs else np.array([], dtype=str)
        return (ids, res, off, recs) if with_records else (ids, res, off)
    n = nrec.value
    res = np.empty(nres.value, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.int64)
    idb, idl = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int32)
    _hip._check(lib, lib.skm_fasta_parse(buf.ctypes.data_as(p), buf.size, threads, n, nres.value, res.ctypes.data_as(p),
                                         off.ctypes.data_as(p), idb.ctypes.data_as(p), idl.ctypes.data_as(p)))
    if n == 0:
        return (np.array([], dtype=str), res, off, None) if with_records else (np.array([], dtype=str), res, off)
    idb, idl = idb[:n], idl[:n]
    width = max(int(idl.max()), 1)
    # ids as a '<U{width}' array: gather the spans into a zero-padded UCS-4 matrix (ASCII bytes are code points)
    cols = np.arange(width, dtype=np.int64)
    take = np.minimum(idb[:, None] + cols[None, :], max(buf.size - 1, 0))
    ids = np.where(cols[None, :] < idl[:, None], buf[take] if buf.size else 0, 0).astype(np.uint32)
    ids = np.ascontiguousarray(ids).view(f"<U{width}").ravel()
    return (ids, res, off, None) if with_records else (ids, res, off)


def read_kmers(filename: str) -> List[str]:
    """One k-mer per line, verbatim order (snekmer/io.py:99-122)."""
    with open(filename) as f:
        return [line.strip() for line in f]


# ``.kmers`` interchange.  rules/kmerize.smk:141-142 pickles a ``snekmer.vectorize.KmerVec``;
# scripts/cluster_cluster.py:53-63 (and model / search) unpickle it with ``snekmer.io.load_pickle``.  A pickle names its
# classes by module path, so a drop-in must WRITE that path: ``dump_kmers`` does (the three classes of
# snekmer/vectorize.py keep their state in ``__dict__`` and define no pickling hook, so the stream is what the reference
# writes), and ``load_pickle`` reads either path, with or without Snekmer installed.
REFERENCE_MODULE = "snekmer.vectorize"
_PICKLED_CLASSES = ("KmerVec", "KmerBasis", "KmerSet")


def _reference_classes():
    """The classes to name in the stream: Snekmer's own when it is importable, otherwise stand-ins that only carry the
    reference's module path and names (never registered in sys.modules: `_StandInPickler` writes their names itself)."""
    try:
        mod = importlib.import_module(REFERENCE_MODULE)
        return {name: getattr(mod, name) for name in _PICKLED_CLASSES}, False
    except Exception:
        pass
    return {name: type(name, (), {"__module__": REFERENCE_MODULE, "__qualname__": name}) for name in _PICKLED_CLASSES}, True


class _StandInPickler(pickle._Pickler):
    """pickle's Python implementation with one change: a stand-in class is written by NAME without the check that
    sys.modules[module].name is that class (the C pickler cannot skip it, which is why round 5 put throw-away modules
    into sys.modules for the duration of a dump, where a concurrent `import snekmer` in another thread could see them)."""

    def __init__(self, file, protocol, standins):
        super().__init__(file, protocol=protocol)
        self._standins = set(standins)

    def save_global(self, obj, name=None):
        if obj not in self._standins:
            return super().save_global(obj, name)
        if self.proto >= 4:
            self.save(obj.__module__)
            self.save(obj.__qualname__)
            self.write(pickle.STACK_GLOBAL)
        else:
            self.write(pickle.GLOBAL + obj.__module__.encode() + b"\n" + obj.__qualname__.encode() + b"\n")
        self.memoize(obj)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


def dump_kmers(obj: Any, file, protocol: int = 4, reference_pickle: bool = True) -> None:
    """``pickle.dump(kmer, f)`` of rules/kmerize.smk:141-142.  With `reference_pickle` (default) the objects of
    snekmer_amd.vectorize are written under the reference's class path ``snekmer.vectorize.*`` with the reference's
    attribute set, so that Snekmer's own ``io.load_pickle`` reads the file unchanged; ``load_pickle`` here reads it too."""
    if not reference_pickle:
        pickle.dump(obj, file, protocol=protocol)
        return
    from . import vectorize as V

    mine = {getattr(V, name): name for name in _PICKLED_CLASSES}
    classes, standins = _reference_classes()

    def convert(o):
        name = mine.get(type(o))
        if name is None:
            return o
        get = getattr(type(o), "__getstate__", None)
        state = get(o) if get is not None and get is not getattr(object, "__getstate__", None) else dict(o.__dict__)
        twin = object.__new__(classes[name])  # same state under the reference's class: pickle then writes what Snekmer writes
        twin.__dict__.update({k: convert(v) for k, v in state.items()})
        return twin

    if standins:
        _StandInPickler(file, protocol, classes.values()).dump(convert(obj))
    else:
        pickle.dump(convert(obj), file, protocol=protocol)


class _KmersUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == REFERENCE_MODULE and name in _PICKLED_CLASSES:
            try:
                return super().find_class(module, name)  # Snekmer is installed: its own classes
            except Exception:
                from . import vectorize as V

                return getattr(V, name)
        return super().find_class(module, name)


def load_pickle(filename: str, mode: str = "rb") -> Any:
    """snekmer/io.py:21-39 (wrapper for ``pickle.load``).  A ``.kmers`` file that names ``snekmer.vectorize.*`` —
    written by Snekmer or by ``dump_kmers`` — loads as snekmer_amd.vectorize objects where Snekmer is not installed."""
    with open(filename, mode) as f:
        return _KmersUnpickler(f).load()


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


def save_npz(filename: str, arrays: Dict[str, np.ndarray], compressed: bool = True, level: int = -1, threads: int = 0,
             _lib=None) -> int:
    """``np.savez_compressed(filename, **arrays)`` (compressed=True: what rules/kmerize.smk:132-139 calls) or
    ``np.savez`` (compressed=False) through the C library's threaded writer (skm_npz_write: host code, no GPU): the same
    zip-of-.npy container, every member deflated in 512 KiB chunks on all cores instead of on one thread.  np.load and
    the reference's io.load_npz (snekmer/io.py:46-96) read the file unchanged.  Like numpy, ".npz" is appended to a
    name without it.  Object arrays (which numpy would pickle) are not supported.  Returns the file's size in bytes."""
    import ctypes as C
    import io as _io
    import os

    from . import _hip

    lib = _lib or _hip.load_library()  # (_lib: the sanitizer build of the host code, tests/asan_driver.py)
    filename = os.fspath(filename)
    if not filename.endswith(".npz"):
        filename += ".npz"
    names, headers, datas = [], [], []
    for name, arr in arrays.items():
        a = np.asanyarray(arr)
        if a.dtype.hasobject:
            raise TypeError(f"save_npz: member {name!r} is an object array")
        if not a.flags.c_contiguous and not a.flags.f_contiguous:
            a = np.ascontiguousarray(a)
        hdr = _io.BytesIO()
        # numpy's own header writer: picks format version 1.0 / 2.0 / 3.0 as np.save does
        np.lib.format._write_array_header(hdr, np.lib.format.header_data_from_array_1_0(a), None)
        names.append(str(name).encode())
        headers.append(hdr.getvalue())
        datas.append(a)
    n = len(names)
    c_names = (C.c_char_p * max(n, 1))(*names)
    hbufs = [C.create_string_buffer(h, len(h)) for h in headers]
    c_headers = (C.c_void_p * max(n, 1))(*[C.addressof(b) for b in hbufs])
    c_hbytes = (C.c_int64 * max(n, 1))(*[len(h) for h in headers])
    c_data = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.nbytes else None for a in datas])
    c_dbytes = (C.c_int64 * max(n, 1))(*[a.nbytes for a in datas])
    size = C.c_int64(0)
    lib.skm_npz_write.restype = C.c_int
    lib.skm_npz_write.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_int64)]
    lib.skm_last_error.restype = C.c_char_p
    _hip._check(lib, lib.skm_npz_write(filename.encode(), n, c_names, c_headers, c_hbytes, c_data, c_dbytes,
                                       (level if compressed else 0), threads, C.byref(size)))
    return int(size.value)


def save_npz_sparse(filename: str, out: Dict[str, np.ndarray], compressed: bool = True) -> None:
    """The rule's ``.npz`` with the count matrix as CSR (`counts_rowptr/col/val` over the columns
    of ``kmerlist``) in place of the dense presence matrix ``vecs``.  `out` is what
    kmerize.vectorize_records returns.  `compressed=False` stores the members as they are; np.load reads both."""
    save_npz(filename, {k: out[k] for k in SPARSE_KEYS}, compressed=compressed)


def load_counts_npz(filename: str):
    """scipy.sparse.csr_matrix of integer k-mer counts [sequences x kmerlist] from a sparse ``.npz``
    (the matrix rules/learn.smk:359-383 and rules/apply.smk:188-206 rebuild in Python)."""
    from scipy.sparse import csr_matrix

    data = np.load(filename)
    n, b = len(data["ids"]), len(data["kmerlist"])
    return csr_matrix((data["counts_val"].astype(np.int64), data["counts_col"].astype(np.int64), data["counts_rowptr"]),
                      shape=(n, b))
