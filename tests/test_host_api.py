"""CPU-side checks: host logic of the reference-shaped API, the C-ABI surface, and the
no-fallback rule.  No kernel is launched here."""
import ctypes
import os
import pickle
import re
import subprocess

import numpy as np
import pytest

from helpers import ensure_red6, gjson

import snekmer_amd as skm
from snekmer_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ensure_red6()


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "snekmer_hip.h")).read()
    declared = set(re.findall(r"\b(skm_[a-z0-9_]+)\s*\(", header))
    declared.discard("skm_ctx")
    lib = _hip.load_library()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_hip.EXPORTED_SYMBOLS) == declared
    assert lib.skm_abi_version() == _hip.ABI_VERSION
    abi_h = int(re.search(r"#define\s+SKM_ABI_VERSION\s+(\d+)", header).group(1))
    assert abi_h == _hip.ABI_VERSION


def test_driver_build_hook_returns():
    """__graft_entry__.build() is what the driver and README call: it must compile (a no-op when up to date),
    load the library and agree with it on the ABI version and the symbol table."""
    import shutil

    import __graft_entry__ as entry

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here: build() cannot compile")
    entry.build()
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert not re.search(r"ABI_VERSION\s*==\s*\d", src), "build() must not pin the ABI to a literal"


def test_product_library_carries_no_result_invalidating_switches():
    """The timing-only ablations live in the -DSKM_DIAG build (libsnekmer_hip_diag.so) alone."""
    blob = open(_hip.LIB_PATH, "rb").read()
    for needle in (b"SKM_COSINE_ABLATE", b"SKM_GRAM_ABLATE", b"SKM_HEAVY_ABLATE", b"skm_debug_gram_phases"):
        assert needle not in blob
    # the ablation switches are read (skm_opts().*_ablate) inside #ifdef SKM_DIAG only, and no call path reads the environment
    for name in ("skm_cosine_csr.hip", "skm_dense.hip", "skm_basis.hip", "skm_kmer.hip", "skm_common.h"):
        src = open(os.path.join(ROOT, "snekmer_amd", "csrc", name)).read()
        assert "getenv(" not in src, name
        for m in re.finditer(r"skm_opts\(\)\.(cosine_ablate|gram_ablate|heavy_ablate|overlap_blocks|dense_split)", src):
            before = src[: m.start()]
            assert before.rfind("#ifdef SKM_DIAG") > before.rfind("#endif"), (name, m.group(1))


def test_options_start_from_the_environment_and_refuse_what_they_do_not_know():
    """skm_set_option / skm_get_option: process-wide switches between exact kernels; the environment is read once."""
    import subprocess
    import sys

    assert _hip.get_option("SKM_SORT") in (None, os.environ.get("SKM_SORT"))
    with _hip.options(SKM_SORT="rocprim", SKM_HEAVY_PANEL=0):
        assert _hip.get_option("SKM_SORT") == "rocprim" and _hip.get_option("SKM_HEAVY_PANEL") == "0"
        with _hip.options(SKM_SORT="onesweep"):
            assert _hip.get_option("SKM_SORT") == "onesweep"
        assert _hip.get_option("SKM_SORT") == "rocprim"
    assert _hip.get_option("SKM_SORT") == os.environ.get("SKM_SORT") and _hip.get_option("SKM_HEAVY_PANEL") == os.environ.get("SKM_HEAVY_PANEL")
    for name, value in (("SKM_SORT", "bubble"), ("SKM_NO_SUCH", "1"), ("SKM_DENSE_VARIANT", "12"), ("SKM_HEAVY_PANEL", "yes"),
                        ("SKM_COSINE_ABLATE", "1")):
        with pytest.raises(_hip.HipError, match="BADARG"):
            _hip.set_option(name, value)
    # a fresh process starts from its environment
    code = "from snekmer_amd import _hip; print(_hip.get_option('SKM_COSINE_PATH'), _hip.get_option('SKM_GRAM_SHAPE'))"
    env = dict(os.environ, SKM_COSINE_PATH="lists", SKM_GRAM_SHAPE="3", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["lists", "3"]


def test_no_gpu_means_loud_failure_not_fallback():
    if _hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_hip.HipUnavailable):
        skm.vectorize.KmerVec("hydro", 4).reduce_vectorize("MKVLAAGIW")
    with pytest.raises(_hip.HipUnavailable):
        skm.vectorize.reduce("MKVL", "hydro")
    with pytest.raises(_hip.HipUnavailable):
        skm.score.connection_matrix_from_features(np.eye(3), metric="cosine")


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "snekmer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} refers to oracle/"


def test_kmerbasis_errors_and_object_shapes_match_reference_fixture():
    g6 = gjson("g6_basis.json")
    kb = skm.vectorize.KmerBasis()
    kb.set_basis(g6["basis"])
    errs = g6["errors"]
    with pytest.raises(TypeError) as e:
        skm.vectorize.KmerBasis().set_basis(5)
    assert [type(e.value).__name__, str(e.value)] == errs["set_basis_type"]
    with pytest.raises(TypeError) as e:
        kb.transform(np.asarray(g6["matrix"]), 5)
    assert [type(e.value).__name__, str(e.value)] == errs["vector_basis_type"]
    with pytest.raises(ValueError) as e:
        kb.transform(np.asarray(g6["matrix"]), g6["vector_basis"][:2])
    assert [type(e.value).__name__, str(e.value)] == errs["shape_mismatch"]
    with pytest.raises(IndexError):
        kb.transform(np.asarray(g6["matrix"])[0], g6["vector_basis"])
    assert errs["one_dim"][0] == "IndexError"
    kv = skm.vectorize.KmerVec("hydro", 3)
    kv.set_kmer_set(["SSS", "SSV", "VVV"])
    assert sorted(kv.__dict__) == g6["kmervec_attrs"]
    assert sorted(kv.kmer_set.__dict__) == g6["kmerset_attrs"]
    assert list(kv.kmer_set.kmers) == g6["kmerset_kmers"]
    kv2 = pickle.loads(pickle.dumps(kv))
    assert kv2.char_set == kv.char_set and kv2.k == 3 and list(kv2.kmer_set.kmers) == g6["kmerset_kmers"]


def test_kmerset_enumeration_and_alphabet_quirks():
    ks = skm.vectorize.KmerSet("hydro", 3)
    assert sorted(ks.kmers) == sorted("".join(p) for p in __import__("itertools").product("SV", repeat=3))
    with pytest.raises(ValueError):
        skm.alphabet.get_alphabet("RED0")
    with pytest.raises(KeyError):
        skm.alphabet.get_alphabet(6)
    assert skm.alphabet.get_alphabet_keys(None) == set(skm.alphabet.StandardAlphabet)
    assert skm.vectorize.KmerVec("red6", 8).char_set == set("ADFKNP")


def test_lut_encode_decode_roundtrip_and_code_order():
    lut = skm.alphabet.build_lut("standard")
    rng = np.random.default_rng(0)
    codes = np.sort(rng.integers(0, 7**12, size=500).astype(np.uint64))
    strings = lut.decode(codes, 12)
    assert list(strings) == sorted(strings)  # numeric order == lexicographic order
    back, ok = lut.encode(list(strings), 12)
    assert ok.all() and (back == codes).all()
    _, ok = lut.encode(["AAAAAAAAAAAX", "AAA"], 12)
    assert not ok.any()
    assert lut.code_bits(8) == 32 and lut.code_bits(12) == 64
    assert skm.alphabet.build_lut("red6").code_bits(12) == 32
    with pytest.raises(ValueError):
        skm.alphabet.build_lut(None).code_bits(15)


def test_utils_contracts():
    from helpers import gnpz

    g = gnpz("g9_connection.npz")
    X = g["X"]
    np.testing.assert_array_equal(skm.utils.to_feature_matrix([list(r) for r in X], np.arange(1, 10)), g["tfm"])
    np.testing.assert_array_equal(skm.utils.to_feature_matrix([list(r) for r in X]), g["tfm_default"])
    assert skm.utils.check_list([1]) and skm.utils.check_list(np.zeros(2)) and not skm.utils.check_list(5)
    data, off = skm.utils.pack_sequences(["MKV", "", "Ä€x"])
    assert off.tolist() == [0, 3, 3, 6] and data[3] == 0xC4 and data[4] == skm.utils.SUBSTITUTE


def test_synth_is_deterministic_and_shaped():
    from snekmer_amd.synth import synth_families

    a = synth_families(1000, 300, seed=5)
    b = synth_families(1000, 300, seed=5)
    assert all((x == y).all() for x, y in zip(a, b))
    res, off, fam = a
    lens = np.diff(off)
    assert set(lens.tolist()) <= {300, 301} and (res[off[1:][lens == 301] - 1] == ord("*")).all()
    assert len(set(fam.tolist())) == 10


def test_sparse_npz_roundtrip(tmp_path):
    """io.save_npz_sparse / load_counts_npz / load_npz on the sparse variant (host only)."""
    out = {
        "kmerlist": np.array(["AAC", "ACA", "CAA"]), "ids": np.array(["s1", "s2", "s3"]), "seqs": np.array(["AACA", "CAAC", "XX"]),
        "lengths": np.array([4, 4, 2]), "counts_rowptr": np.array([0, 2, 4, 4], dtype=np.int64),
        "counts_col": np.array([0, 1, 0, 2], dtype=np.uint32), "counts_val": np.array([1, 1, 2, 1], dtype=np.uint32),
    }
    path = str(tmp_path / "x.npz")
    skm.io.save_npz_sparse(path, out)
    C = skm.io.load_counts_npz(path)
    assert C.shape == (3, 3) and C.toarray().tolist() == [[1, 1, 0], [2, 0, 1], [0, 0, 0]]
    (kl,), df = skm.io.load_npz(path)
    assert list(kl) == ["AAC", "ACA", "CAA"] and list(df["sequence_length"]) == [4, 4, 2]
    assert [v.tolist() for v in df["sequence_vector"]] == [[1.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 0.0, 0.0]]


def test_reference_written_kmers_pickle_loads_through_the_module_alias():
    """G14: a .kmers pickle written by the real reference (rules/kmerize.smk:141-142) is read with snekmer_amd
    after aliasing the module path, as INTEGRATION.md describes; and a pickle written here carries exactly the
    reference's attribute set (the cached LUT stays out of it)."""
    import sys

    g = gjson("g14_reference_kmers_pickle.json")
    saved = {k: sys.modules.get(k) for k in ("snekmer", "snekmer.vectorize")}
    try:
        sys.modules["snekmer"] = skm
        sys.modules["snekmer.vectorize"] = skm.vectorize
        obj = pickle.loads(bytes.fromhex(g["pickle_hex"]))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert isinstance(obj, skm.vectorize.KmerVec) and isinstance(obj.basis, skm.vectorize.KmerBasis)
    assert (obj.alphabet, obj.k, sorted(obj.char_set), obj.snekmer_version) == (g["alphabet"], g["k"], g["char_set"], g["snekmer_version"])
    assert sorted(obj.__dict__.keys()) == g["attrs"]
    assert [str(x) for x in list(obj.kmer_set.kmers)[:5]] == g["first"] and len(list(obj.kmer_set.kmers)) == g["n_kmers"]
    mine = skm.vectorize.KmerVec("hydro", 14)
    mine.set_kmer_set(list(obj.kmer_set.kmers))
    mine._lut()  # populate the cache: it must not reach the pickle
    state = pickle.loads(pickle.dumps(mine)).__dict__
    assert sorted(state.keys()) == g["attrs"]


def test_kmers_written_here_is_the_stream_the_reference_writes(tmp_path):
    """The reverse direction of G14: a .kmers file written by snekmer_amd (io.dump_kmers, what vectorize_fasta calls) names
    ``snekmer.vectorize.KmerVec / KmerBasis / KmerSet`` and is, opcode for opcode, the stream the real reference wrote for the same
    k-mer list (fixture G14) - so snekmer/io.py:21-39 + scripts/cluster_cluster.py:53-63 read it unchanged, with no snekmer_amd
    installed.  (tests/golden/make_golden.py additionally unpickles such a file with the imported reference in the build
    container and records the outcome in the fixture.)  Nothing named snekmer is left in sys.modules, and io.load_pickle reads
    the file back as this package's classes."""
    import io as _io
    import pickletools
    import sys

    g = gjson("g14_reference_kmers_pickle.json")
    ref_blob = bytes.fromhex(g["pickle_hex"])
    (tmp_path / "ref.kmers").write_bytes(ref_blob)
    ref_obj = skm.io.load_pickle(str(tmp_path / "ref.kmers"))  # the reference's own file, no alias needed
    assert type(ref_obj) is skm.vectorize.KmerVec
    mine = skm.vectorize.KmerVec(g["alphabet"], g["k"])
    mine.set_kmer_set(ref_obj.kmer_set._kmerlist)
    mine.snekmer_version = g["snekmer_version"]
    mine._lut()
    had = "snekmer" in sys.modules
    buf = _io.BytesIO()
    skm.io.dump_kmers(mine, buf)
    assert ("snekmer" in sys.modules) == had and ("snekmer.vectorize" in sys.modules) == had
    blob = buf.getvalue()

    def ops(b):  # opcode stream without the frame sizes; the 2-element char_set is a set: its order follows the hash seed
        out = []
        for op, arg, _ in pickletools.genops(b):
            if op.name == "FRAME":
                continue
            out.append((op.name, "<class letter>" if isinstance(arg, str) and len(arg) == 1 else
                        (arg if not isinstance(arg, bytes) or len(arg) < 64 else hash(arg))))
        return out

    assert ops(blob) == ops(ref_blob)
    assert len(blob) == len(ref_blob)
    globals_named = [a for name, a in ops(blob) if isinstance(a, str) and (a.startswith("snekmer") or a.startswith("Kmer"))]
    assert globals_named[:3] == ["snekmer.vectorize", "KmerVec", "KmerBasis"] and "KmerSet" in globals_named
    assert not any("snekmer_amd" in str(a) for _, a in ops(blob))
    # the flag off: this package's class path (a file for snekmer_amd only)
    buf2 = _io.BytesIO()
    skm.io.dump_kmers(mine, buf2, reference_pickle=False)
    assert b"snekmer_amd.vectorize" in buf2.getvalue()
    (tmp_path / "mine.kmers").write_bytes(blob)
    back = skm.io.load_pickle(str(tmp_path / "mine.kmers"))
    assert type(back) is skm.vectorize.KmerVec and type(back.basis) is skm.vectorize.KmerBasis and type(back.kmer_set) is skm.vectorize.KmerSet
    assert sorted(back.__dict__) == g["attrs"] and (back.kmer_set._kmerlist == ref_obj.kmer_set._kmerlist).all()
    if "reverse_direction" in g:  # recorded by make_golden.py with the real reference imported
        assert g["reverse_direction"]["loaded_class"] == "snekmer.vectorize.KmerVec" and g["reverse_direction"]["state_equal"]


def test_threaded_npz_writer_matches_numpy_savez(tmp_path):
    """skm_npz_write (host code) against np.savez_compressed / np.savez on the rule's members (rules/kmerize.smk:132-139):
    the same member names, dtypes, shapes and values through np.load and through the reference-shaped io.load_npz, a valid
    zip (CRCs checked by zipfile), and a size close to numpy's own (the '<U' members go through the library's own
    deflate encoder, 5-7x faster than zlib level 6 per thread and 4-13 % larger on k-mer lists; numeric members that do
    not compress beyond entropy coding are Huffman-coded only)."""
    import zipfile

    from snekmer_amd import io as sio

    rng = np.random.default_rng(5)
    n, b = 300, 4000
    arrays = dict(
        kmerlist=np.array(["".join(rng.choice(list("ADFKNP"), 12)) for _ in range(b)], dtype=str),
        ids=np.array([f"seq{i}" for i in range(n)], dtype=str),
        seqs=np.array(["".join(rng.choice(list("ADFKNPX"), int(rng.integers(0, 400)))) for _ in range(n)], dtype=str),
        vecs=(rng.random((n, b)) < 0.03).astype(np.float64),
        lengths=rng.integers(0, 400, size=n),
    )
    ref = tmp_path / "ref.npz"
    np.savez_compressed(ref, **arrays)
    for compressed, threads in ((True, 0), (True, 1), (True, 5), (False, 0)):
        path = tmp_path / f"mine_{int(compressed)}_{threads}"
        size = sio.save_npz(str(path), arrays, compressed=compressed, threads=threads)  # ".npz" is appended, as numpy does
        path = str(path) + ".npz"
        assert size == os.path.getsize(path)
        with zipfile.ZipFile(path) as z:
            assert z.testzip() is None
            assert [i.filename for i in z.infolist()] == [k + ".npy" for k in arrays]
            assert all(i.compress_type == (zipfile.ZIP_DEFLATED if compressed else zipfile.ZIP_STORED) for i in z.infolist())
        got, want = np.load(path), np.load(ref)
        for key in arrays:
            assert got[key].dtype == want[key].dtype and got[key].shape == want[key].shape and (got[key] == want[key]).all()
        if compressed:
            assert size <= 1.15 * os.path.getsize(ref)
        (kmerlist,), df = sio.load_npz(path)
        assert list(kmerlist) == list(arrays["kmerlist"]) and list(df["sequence_id"]) == list(arrays["ids"])
        assert (np.asarray(list(df["sequence_vector"])) == arrays["vecs"]).all()
    # the string encoder on everything a '<U' array can hold: characters beyond latin-1 and beyond the BMP, empty and
    # ragged strings, 2-D, width 1, items longer than the 32 KiB window and longer than a chunk, sorted k-mers of
    # every size from none to several chunks; big-endian strings take the zlib path
    words = ["", "a", "αβγδ", "日本語テキスト", "😀😀x", "A" * 999, "z" * 5, "mixedΩ😀"]
    arr = np.array([words[i] for i in rng.integers(0, len(words), size=3000)])
    cases = [{"s": arr, "t": arr.reshape(30, 100), "one": np.array(list("ACDEFGHIKLMNPQRSTVWY" * 50)), "e": np.array([""] * 7),
              "num": np.arange(100000), "be": arr[:200].astype(">U999")},
             {"wide": np.array(["".join(chr(c) for c in rng.integers(65, 91, size=int(L))) for L in rng.integers(0, 12000, size=120)])},
             {"huge": np.array(["Q" * 600000, "R" * 10])}]
    for nk in (0, 1, 2, 5, 1000, 120000):
        codes = np.unique(rng.integers(0, 6**12, size=nk, dtype=np.int64)) if nk else np.zeros(0, np.int64)
        digits = (codes[:, None] // (6 ** np.arange(11, -1, -1))) % 6
        cases.append({"kmerlist": np.frombuffer(b"SNDQEH", dtype=np.uint8)[digits].astype(np.uint8).view("S12").reshape(-1).astype("<U12")
                      if nk else np.zeros(0, "<U12")})
    for i, members in enumerate(cases):
        path = str(tmp_path / f"strings_{i}.npz")
        sio.save_npz(path, members, threads=3)
        with zipfile.ZipFile(path) as z:
            assert z.testzip() is None
        with np.load(path) as got:
            for key, want in members.items():
                assert got[key].dtype == want.dtype and got[key].shape == want.shape and (got[key] == want).all(), (i, key)
    # integer members take the same unit encoder (no item structure): column ids with repeated stretches, counts, row
    # starts (8-byte elements that differ in the low byte), negative values, sizes around the chunk size
    ids = np.sort(rng.integers(0, 90000, size=(700, 289)), axis=1).astype(np.uint32)
    ids[1::3] = ids[0::3][: len(ids[1::3])]  # every third row repeats its neighbour, like a relative's row of k-mer ids
    numbers = {"col": ids.reshape(-1), "val": (1 + (rng.random(300000) < 0.02)).astype(np.uint32), "rowptr": np.cumsum(rng.integers(200, 300, size=40001)),
               "neg": rng.integers(-5, 5, size=(1 << 17) + 3).astype(np.int32), "edge": np.arange((1 << 19) // 8 + 1, dtype=np.int64),
               "u8": rng.integers(0, 2**63, size=70000, dtype=np.uint64), "f4": rng.normal(size=3000).astype(np.float32)}
    path = str(tmp_path / "numbers.npz")
    size = sio.save_npz(path, numbers, threads=4)
    np.savez_compressed(tmp_path / "numbers_ref.npz", **numbers)
    assert size <= 1.10 * os.path.getsize(tmp_path / "numbers_ref.npz")
    with zipfile.ZipFile(path) as z:
        assert z.testzip() is None
    with np.load(path) as got:
        for key, want in numbers.items():
            assert got[key].dtype == want.dtype and (got[key] == want).all(), key
    # units that do not compress at all (random 32-bit patterns viewed as '<U1'): the encoder falls back to stored blocks
    for count in (16, 1000, 300001):
        raw = rng.integers(0, 2**32, size=count, dtype=np.uint32)
        path = str(tmp_path / f"noise_{count}.npz")
        size = sio.save_npz(path, {"a": raw.view("<U1")}, threads=3)
        with zipfile.ZipFile(path) as z:
            assert z.testzip() is None and z.read("a.npy")[-4 * count:] == raw.tobytes()
        assert size <= 4 * count + 600
    with pytest.raises(TypeError):
        sio.save_npz(str(tmp_path / "obj"), {"o": np.array([{}], dtype=object)})
    with pytest.raises(_hip.HipError):
        sio.save_npz(str(tmp_path / "no" / "dir" / "x.npz"), arrays)


def test_runtime_resources_are_only_taken_in_the_pool_module():
    """Design rule of round 6: device memory, streams and events are taken from the HIP runtime in
    ONE translation unit (csrc/skm_mem.hip) and recycled; no other source calls hipMalloc / hipFree / hipStreamCreate* /
    hipStreamDestroy / hipEventCreate* / hipEventDestroy (three long fuzz runs of round 5 stopped inside a hipFree that met a
    busy device).  Pinned staging memory of the per-record API (skm_host_alloc) is the one exception, in skm_api.hip."""
    import glob
    import re

    csrc = os.path.join(ROOT, "snekmer_amd", "csrc")
    pat = re.compile(r"\b(hipMalloc|hipFree|hipStreamCreate\w*|hipExtStreamCreate\w*|hipStreamDestroy|hipEventCreate\w*|hipEventDestroy|hipMallocAsync|hipFreeAsync)\s*\(")
    bad = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))):
        if os.path.basename(path) == "skm_mem.hip":
            continue
        for no, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if pat.search(code):
                bad.append(f"{os.path.basename(path)}:{no}: {line.strip()}")
    assert not bad, bad
