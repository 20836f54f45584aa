"""The C library's threaded FASTA reader (skm_fasta_index / skm_fasta_parse: host code, runs without a GPU) against
the text-mode reader, which restates Biopython's SimpleFastaParser — the reader behind the `SeqIO.parse(fasta, "fasta")`
loops of snekmer/rules/kmerize.smk:90-129 (Biopython itself is not installed here)."""
import numpy as np
import pytest

from snekmer_amd import io
from snekmer_amd.utils import pack_sequences

HAND = [
    b"", b"\n\n", b">only", b">x\n", b"ACGT\n", b">\n", b">\nAA\n>\nBB",
    b"junk before\nmore\n>a desc\nMKV\nLAA\n>b\n\n>c\tx\nAB CD\t \n  EF\r\nGH\rIJ\n>\nXX\n> lead\nYY",
    b">a\r\nMK V\r\n>b\rLL\r", b">a b c\n\tMKV \x0b\n\x1c>notheader\nZZ*\n", b">a\n*\n>b\nMKV**\n",
    b">sp|P12345|NAME_ORG desc\nMKVLAAGIWSTC\nDEFHNPQRY\n",
]


def _check(tmp_path, blob, threads):
    path = tmp_path / "x.fa"
    path.write_bytes(blob)
    recs = io.read_fasta(str(path))
    want_res, want_off = pack_sequences([s for _, s in recs])
    for th in threads:
        ids, res, off = io.read_fasta_packed(str(path), threads=th)
        assert list(ids) == [r[0] for r in recs]
        assert ids.dtype.kind == "U" and (off == want_off).all() and (res == want_res).all()


def test_hand_cases_and_text_mode_semantics(tmp_path):
    for blob in HAND:
        _check(tmp_path, blob, (1, 0))
    path = tmp_path / "y.fa"
    path.write_bytes(HAND[7])
    assert io.read_fasta(str(path)) == [("a", "MKVLAA"), ("b", ""), ("c", "ABCDEFGHIJ"), ("", "XX"), ("lead", "YY")]


def test_random_byte_soup(tmp_path):
    rng = np.random.default_rng(0)
    alphabet = list(b"ACDEFGHIKLMNPQRSTVWY*X \t\r\n\n\n>>> ab\x0b\x0c\x1c")
    for _ in range(200):
        blob = bytes(rng.choice(alphabet, size=int(rng.integers(0, 400))).tolist())
        _check(tmp_path, blob, (1, 3))


def test_multi_megabyte_file_every_thread_count(tmp_path):
    """Chunks are cut at line starts; records that span a cut, text before the first header that fills whole chunks,
    mixed line endings."""
    rng = np.random.default_rng(1)
    aa = list(b"ACDEFGHIKLMNPQRSTVWY")
    parts = [b"pre\n" * 300000]
    for i in range(12000):
        parts.append(b">seq%d some description\n" % i)
        s = bytes(rng.choice(aa, size=int(rng.integers(0, 900))).tolist())
        eol = (b"\r\n", b"\n", b"\r")[i % 3]
        for j in range(0, len(s), 60):
            parts.append(s[j:j + 60] + eol)
    _check(tmp_path, b"".join(parts), (1, 2, 5, 8, 0))


def test_non_ascii_takes_the_text_path(tmp_path):
    blob = ">aä d\nMKÄVLAα\n>b\nMKV\n".encode("utf-8")
    path = tmp_path / "u.fa"
    path.write_bytes(blob)
    ids, res, off = io.read_fasta_packed(str(path))
    recs = io.read_fasta(str(path))
    assert list(ids) == [r[0] for r in recs] == ["aä", "b"]
    want_res, want_off = pack_sequences([s for _, s in recs])
    assert (res == want_res).all() and (off == want_off).all()


def test_size_mismatch_is_an_error(tmp_path):
    import ctypes as C

    from snekmer_amd import _hip

    lib = _hip.load_library()
    buf = np.frombuffer(b">a\nMKV\n", dtype=np.uint8)
    p = C.c_void_p
    res, off, idb, idl = np.zeros(8, np.uint8), np.zeros(3, np.int64), np.zeros(2, np.int64), np.zeros(2, np.int32)
    rc = lib.skm_fasta_parse(buf.ctypes.data_as(p), buf.size, 1, 2, 3, res.ctypes.data_as(p), off.ctypes.data_as(p),
                             idb.ctypes.data_as(p), idl.ctypes.data_as(p))
    assert rc == -1 and b"records" in lib.skm_last_error()
