"""Pins the oracle (oracle/ref_path.py + oracle/kmer_oracle.c) and the host alphabet tables
against fixtures generated from the imported reference (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from helpers import ALL_ALPHABETS, GOLDEN, alpha_key, csr_to_dense, demo_records, ensure_red6, gjson, gnpz, parse_tag
from oracle import c_oracle, ref_path
from snekmer_amd import alphabet as A
from snekmer_amd.utils import pack_sequences, unpack_sequences

ensure_red6()


def test_g1_alphabet_tables():
    g1 = gjson("g1_alphabets.json")
    meta = g1.pop("_meta")
    for name, ref in g1.items():
        key = alpha_key(name)
        assert A.FULL_ALPHABETS[name] == ref["full"]
        assert sorted(A.get_alphabet_keys(key)) == ref["char_set"]
        assert A.get_alphabet(key) == ref["short"]
        lut = A.build_lut(key)
        assert lut.letters == "".join(ref["char_set"])
        for b in range(256):
            ch = chr(b)
            t = ref["full"].get(ch, ch)
            assert chr(lut.translate[b]) == t
            assert lut.rank[b] == (ref["char_set"].index(t) if t in ref["char_set"] else 0xFF)
    assert {str(k): v for k, v in A.ALPHABET_ORDER.items()} == meta["ALPHABET_ORDER"]
    assert A.ALPHABET2ID == meta["ALPHABET2ID"]
    assert A.ALPHABET_ID == meta["ALPHABET_ID"]
    with pytest.raises(ValueError) as e:
        A.check_valid("no-such-alphabet")
    assert str(e.value) == meta["check_valid_error"]


def test_g2_edge_cases_python_and_c_oracle():
    g2 = gjson("g2_edge_cases.json")
    assert len(g2) > 500
    for case in g2:
        key = alpha_key(case["alphabet"])
        table = A.FULL_ALPHABETS[case["alphabet"]]
        k = case["k"]
        assert ref_path.reduce(case["seq"], table) == case["reduced"]
        got = ref_path.reduce_vectorize(case["seq"], k, table)
        assert list(got) == case["kmers"]
        assert str(got.dtype) == case["dtype"] and list(got.shape) == case["shape"]
        # C oracle on the packed form
        lut = A.build_lut(key)
        if lut.nsym**k >= 2**64:
            continue
        seq, off = pack_sequences([case["seq"]])
        out, outlen = c_oracle.recode(lut.translate, seq, off)
        red = unpack_sequences(out, off, outlen)[0]
        if all(ord(c) < 256 for c in case["seq"]):
            assert red == case["reduced"]
        codes, nwin = c_oracle.kmer_codes(lut.rank, lut.nsym, k, seq, off)
        w = int(nwin[0])
        assert w == max(0, len(case["reduced"]) - k + 1)
        valid = codes[:w][codes[:w] != np.iinfo(np.uint64).max]
        assert list(lut.decode(valid, k)) == case["kmers"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g3_demo_*_mf*.npz"))))
def test_g3_g4_g5_demo(path):
    tag = os.path.basename(path)[len("g3_demo_") : -4]
    alphabet, k, mf = parse_tag(tag)
    name = "None" if alphabet is None else alphabet
    g = np.load(path)
    recs = demo_records()
    table = A.FULL_ALPHABETS[name]
    out = ref_path.kmerize_rule(recs, k, table, min_filter=mf)
    vecs = np.unpackbits(g["vecs_bits"], axis=1)[:, : g["vecs_shape"][1]]
    assert list(out["kmerlist"]) == list(g["kmerlist"])
    assert list(out["ids"]) == list(g["ids"]) and list(out["seqs"]) == list(g["seqs"])
    assert list(out["lengths"]) == list(g["lengths"])
    assert (out["vecs"] == vecs).all()
    counts, _ = ref_path.count_matrix(out["seqs"], out["kmerlist"])
    gold_counts = csr_to_dense(g["counts_rowptr"], g["counts_col"], g["counts_val"], len(g["kmerlist"]))
    assert (counts == gold_counts).all()
    np.testing.assert_allclose(ref_path.cosine_similarity(counts), g["cosine"], atol=1e-13)
    np.testing.assert_allclose(ref_path.cosine_similarity(g["totals"], counts).T, g["cosine_rect"], atol=1e-13)

    # C oracle: CSR counts, basis in first-seen order with min_filter, cosine rows
    lut = A.build_lut(alphabet)
    seq, off = pack_sequences([s for _, s in recs])
    rowptr, codes, cnt, first = c_oracle.count_csr(lut.rank, lut.nsym, k, seq, off)
    basis, df, tot, fk, col = c_oracle.basis(rowptr, codes, cnt, first)
    order = np.argsort(fk, kind="stable")
    keep = order[tot[order] > mf]
    assert list(lut.decode(basis[keep], k)) == list(g["kmerlist"])
    newcol = np.full(len(basis), -1, dtype=np.int64)
    newcol[keep] = np.arange(len(keep))
    dense = np.zeros((len(recs), len(keep)), dtype=np.int64)
    for i in range(len(recs)):
        s, e = rowptr[i], rowptr[i + 1]
        c = newcol[col[s:e]]
        dense[i, c[c >= 0]] = cnt[s:e][c >= 0]
    assert (dense == gold_counts).all()
    if mf == 0:
        S = c_oracle.cosine_rows(rowptr, col, cnt, len(basis), np.arange(len(recs)))
        np.testing.assert_allclose(S, g["cosine"], atol=1e-12)


def test_g3_basis_file_branch():
    g = gnpz("g3_demo_hydro_k14_basisfile.npz")
    recs = demo_records()
    out = ref_path.kmerize_rule(recs, 14, A.FULL_ALPHABETS["hydro"], basis=list(g["kmerlist"]))
    vecs = np.unpackbits(g["vecs_bits"], axis=1)[:, : g["vecs_shape"][1]]
    assert (out["vecs"] == vecs).all()
    counts, _ = ref_path.count_matrix(out["seqs"], out["kmerlist"])
    gold = csr_to_dense(g["counts_rowptr"], g["counts_col"], g["counts_val"], len(g["kmerlist"]))
    assert (counts == gold).all()


def test_g6_basis_transform():
    g6 = gjson("g6_basis.json")
    got = ref_path.basis_transform(g6["basis"], np.asarray(g6["matrix"]), g6["vector_basis"])
    assert got.tolist() == g6["out"]


def test_g7_make_feature_matrix():
    for case in gjson("g7_feature_matrix.json"):
        rows, kl = ref_path.make_feature_matrix([np.asarray(v, dtype=str) for v in case["vecs"]], case["min_filter"])
        assert [str(x) for x in kl] == case["kmerlist"]
        assert [r.tolist() for r in rows] == case["rows"]


@pytest.mark.parametrize("name,k", [("red6", 12), ("standard", 12), ("hydro", 20)])
def test_g8_synthetic(name, k):
    from snekmer_amd.synth import synth_families

    g = gnpz(f"g8_synth_{name}_k{k}.npz")
    idx = 5 if name == "hydro" else 2
    res, off, fam = synth_families(256, 300, family=16, seed=20250523 + idx)
    assert (res == g["residues"]).all() and (off == g["offsets"]).all() and (fam == g["family"]).all()
    lut = A.build_lut(name)
    rowptr, codes, cnt, first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
    basis, df, tot, fk, col = c_oracle.basis(rowptr, codes, cnt, first)
    order = np.argsort(fk, kind="stable")
    assert list(lut.decode(basis[order], k)) == list(g["kmerlist"])
    rank_of = np.empty(len(basis), dtype=np.int64)
    rank_of[order] = np.arange(len(basis))
    n = len(off) - 1
    for i in range(0, n, 17):
        s, e = rowptr[i], rowptr[i + 1]
        gs, ge = g["counts_rowptr"][i], g["counts_rowptr"][i + 1]
        mine = dict(zip(rank_of[col[s:e]].tolist(), cnt[s:e].tolist()))
        gold = dict(zip(g["counts_col"][gs:ge].tolist(), g["counts_val"][gs:ge].tolist()))
        assert mine == gold
    S = c_oracle.cosine_rows(rowptr, col, cnt, len(basis), np.arange(n))
    np.testing.assert_allclose(S, g["cosine"], atol=1e-12)


def test_g9_connection_matrix():
    g = gnpz("g9_connection.npz")
    np.testing.assert_allclose(ref_path.cosine_distances(g["X"]), g["cosine"], atol=1e-13)
    np.testing.assert_allclose(ref_path.hamming_similarity(g["X"] > 0), g["jaccard"], atol=1e-13)


def test_g15_hamming_and_jaccard_on_count_and_real_matrices():
    """G15: the reference's default connection_matrix_from_features(counts) and scipy's Jaccard distance on
    non-binary matrices, against the oracle's restatements."""
    from helpers import csr_to_dense

    g, g15 = gnpz("g9_connection.npz"), gnpz("g15_hamming_counts.npz")
    g3 = gnpz("g3_demo_hydro_k14_mf0.npz")
    counts = csr_to_dense(g3["counts_rowptr"], g3["counts_col"], g3["counts_val"], len(g3["kmerlist"]))
    for M, hk, jk in ((g["X"], "default_x", "jaccard_x"), (counts, "default_demo_counts", "jaccard_demo_counts"),
                      (g15["F"], "default_float", "jaccard_float")):
        np.testing.assert_allclose(ref_path.hamming_similarity(M), g15[hk], atol=1e-15)
        np.testing.assert_allclose(ref_path.jaccard_distance(M), g15[jk], atol=1e-15)


def test_g10_real_proteome_oracle():
    """The reference's own CI proteome (.test/input_learnapp, k=8, alphabet 2 = solvacc)."""
    g = gnpz("g10_proteome_solvacc_k8.npz")
    recs = ref_path.read_fasta(os.path.join(GOLDEN, "data", "UP000322080_2603819.fasta"))
    assert [r[0] for r in recs] == list(g["ids"]) and [len(r[1]) for r in recs] == list(g["lengths"])
    lut = A.build_lut(2)
    seq, off = pack_sequences([s for _, s in recs])
    rowptr, codes, cnt, first = c_oracle.count_csr(lut.rank, lut.nsym, 8, seq, off)
    basis, df, tot, fk, col = c_oracle.basis(rowptr, codes, cnt, first)
    order = np.argsort(fk, kind="stable")
    assert list(lut.decode(basis[order], 8)) == list(g["kmerlist"])
    assert (tot[order] == g["col_totals"]).all() and (df[order] == g["col_df"]).all()
    assert len(codes) == int(g["nnz"][0]) and int(cnt.max()) == int(g["max_count"][0])
    assert (np.add.reduceat(np.r_[cnt, 0].astype(np.int64), rowptr[:-1]) * (np.diff(rowptr) > 0) == g["row_count_sums"]).all()
    assert (np.diff(rowptr) == g["row_presence_sums"]).all()
    S = c_oracle.cosine_rows(rowptr, col, cnt, len(basis), g["sample_rows"])
    np.testing.assert_allclose(S, g["cosine_rows"], atol=1e-12)


def test_g11_jaccard_distance_formula():
    """The set formula the device epilogue uses equals scipy's pdist(X, 'jaccard') on the fixture."""
    g3 = gnpz("g3_demo_hydro_k14_mf0.npz")
    vecs = np.unpackbits(g3["vecs_bits"], axis=1)[:, : g3["vecs_shape"][1]].astype(np.int64)
    inter = vecs @ vecs.T
    sizes = vecs.sum(axis=1)
    uni = sizes[:, None] + sizes[None, :] - inter
    D = np.where(uni > 0, (uni - inter) / np.maximum(uni, 1), 0.0)
    np.testing.assert_allclose(D, gnpz("g11_jaccard_demo_hydro_k14.npz")["jaccard_distance"], atol=1e-12)


@pytest.mark.parametrize("tag,name,k", [("standard_k12", "standard", 12), ("hydro_k14", "hydro", 14), ("solvacc_k8", "solvacc", 8)])
def test_g12_apply_epilogue(tag, name, k):
    """The apply epilogue restatement (oracle/ref_path.py) against rules/apply.smk:278-328 run with the
    real sklearn / pandas in the build container."""
    import io

    import pandas as pd

    g = gnpz(f"g12_apply_{tag}.npz")
    lut = A.build_lut(name)
    res, off, fam, train = g["residues"], g["offsets"], g["family"], g["train"]
    rowptr, codes, cnt, first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
    basis, df, tot, fk, col = c_oracle.basis(rowptr, codes, cnt, first)
    counts = csr_to_dense(rowptr, col, cnt, len(basis))
    nfam = int(fam.max()) + 1
    totals = np.vstack([counts[train & (fam == f)].sum(axis=0) for f in range(nfam)])
    conf = pd.read_csv(io.StringIO(str(g["confidence_csv"])))
    table = dict(zip(conf.iloc[:, 0].to_numpy(), conf.iloc[:, 1].to_numpy()))
    out = ref_path.apply_epilogue(totals, counts, list(g["names"]), table)
    assert (out["sorted_vals"] == g["sorted_vals"]).all()
    np.testing.assert_allclose(out["score_rank"], g["score_rank"], atol=1e-13)
    assert (out["delta"] == g["delta"]).all()
    assert (out["Prediction"] == g["Prediction"]).all()
    nan = np.isnan(g["Confidence"])
    assert (np.isnan(out["Confidence"]) == nan).all() and (out["Confidence"][~nan] == g["Confidence"][~nan]).all()
    assert nan.sum() > 0  # the table has holes on purpose


def test_g13_float_features_restatement():
    g = gnpz("g13_float_features.npz")
    np.testing.assert_allclose(ref_path.cosine_similarity(g["X"]), g["cos_xx"], atol=1e-14)
    np.testing.assert_allclose(ref_path.cosine_similarity(g["X"], g["Y"]), g["cos_xy"], atol=1e-14)
    np.testing.assert_allclose(ref_path.cosine_distances(g["X"]), g["conn_cosine_x"], atol=1e-14)


def test_multithreaded_oracle_forms_equal_the_pinned_single_threaded_ones():
    """orc_*_mt (OpenMP) are the same restatements spread over rows / code buckets: identical output."""
    from snekmer_amd.synth import synth_families

    for name, k, seed in (("red6", 12, 3), ("standard", 12, 4), ("hydro", 20, 5)):
        lut = A.build_lut(name)
        res, off, _ = synth_families(700, 300, family=25, seed=seed)
        one = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
        many = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off, threads=0)
        assert all((a == b).all() for a, b in zip(one, many))
        b1 = c_oracle.basis(*one)
        bm = c_oracle.basis(*one, threads=0)
        assert all((a == b).all() for a, b in zip(b1, bm))
        rowptr, codes, cnt, _ = one
        col = b1[4]
        n = len(off) - 1
        ref = c_oracle.cosine_rows(rowptr, col, cnt, len(b1[0]), np.arange(n))
        total, S = c_oracle.cosine_all(rowptr, col, cnt, len(b1[0]), keep=True)
        assert np.abs(S - ref).max() <= 1e-6 and abs(total - float(S.sum(dtype=np.float64))) <= 1e-6 * max(1.0, total)
        total2, rowsum, rownnz = c_oracle.cosine_all(rowptr, col, cnt, len(b1[0]), stats=True)
        assert np.abs(rowsum - S.sum(axis=1, dtype=np.float64)).max() <= 1e-9 and (rownnz == (S != 0).sum(axis=1)).all()
        rows = np.arange(0, n, 37)
        G = c_oracle.sampled_gram(rowptr, codes, cnt, rows)
        nsq = np.add.reduceat(cnt.astype(np.float64) ** 2, rowptr[:-1])
        nsq[np.diff(rowptr) == 0] = 1.0
        nr = np.sqrt(nsq)
        assert np.abs(G / nr[rows][:, None] / nr[None, :] - ref[rows]).max() <= 1e-12
