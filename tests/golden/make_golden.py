#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the build container (the reference tree is not on the GPU box):

    python tests/golden/make_golden.py /root/reference

It imports the reference's own ``snekmer.alphabet`` / ``snekmer.vectorize`` / ``snekmer.score``
modules through an empty stub package (``import snekmer`` itself needs seaborn/hdbscan, which
are absent) and scikit-learn's ``cosine_similarity`` (the reference's
un-vendored dependency), feeds them the inputs below and stores inputs + outputs.  The
Snakemake rule bodies cannot be imported (no snakemake), so the small loops that surround
the imported calls in ``rules/kmerize.smk:89-129`` and ``rules/learn.smk:359-383`` are
re-expressed here around the *imported* ``KmerVec.reduce_vectorize`` / ``reduce``.

Fixtures are data only: inputs and expected outputs.
"""
import json
import os
import shutil
import sys
import types
from collections import Counter

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def import_reference(ref_root):
    pkg = types.ModuleType("snekmer")
    pkg.__path__ = [os.path.join(ref_root, "snekmer")]
    sys.modules["snekmer"] = pkg
    import snekmer.alphabet as alphabet  # noqa
    import snekmer.vectorize as vectorize  # noqa
    import snekmer.score as score  # noqa
    import snekmer.utils as utils  # noqa

    return alphabet, vectorize, score, utils


RED6 = {"AGILMV": "A", "PH": "P", "FWY": "F", "NQSTC": "N", "DE": "D", "KR": "K", "_keys": "APFNDK"}

EDGE_STRINGS = [
    "",
    "MKV",
    "mkvlaagi",
    "MKVL*AGIW**",
    "MKVLXAGIWST",
    "MKBBPG*",
    "MKEEKNR",
    "MKVLAAGIWSTXAAAA*",
    "***",
    "*MKVLAAGIW",
    "MKVLAAGIWSTCDEFHNPQRY",
    "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
    "MK-_!VLA^#$@.%&GIW",
    "MKVL AAGIW\tSTC",
    "MKUVLOAAGIWZSTJC",
    "MKVLAAGIWSTCDEFHNPQRYMKVLAAGIWSTCDEFHNPQRYMKVLAAGIWSTCDEFHNPQRY*",
    "MKÄVLAAGIWαSTC",
]
EDGE_KS = [1, 3, 8, 12, 14, 20]


def read_fasta(path):
    recs, name, chunks = [], None, []
    for line in open(path):
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(chunks)))
            name, chunks = line[1:].split()[0], []
        else:
            chunks.append(line.strip())
    if name is not None:
        recs.append((name, "".join(chunks)))
    return recs


def run_rule(V, A, records, alphabet, k, min_filter=0, basis=None):
    """Drive the imported reference objects the way rules/kmerize.smk:67-139 does."""
    kmer = V.KmerVec(alphabet=alphabet, k=k)
    if basis is None:
        tally = {}
        for _, seq in records:
            for key in kmer.reduce_vectorize(seq):
                tally[key] = tally.get(key, 0) + 1
        names = np.array(list(tally.keys()))
        keep = np.array(list(tally.values())) > min_filter
        kmerbasis = names[keep] if len(names) else names
    else:
        kmerbasis = list(basis)
    kmer.set_kmer_set(kmerbasis)
    vecs = np.zeros((len(records), len(kmerbasis)))
    seqs, ids, lengths = [], [], []
    for n, (rid, seq) in enumerate(records):
        present = kmer.reduce_vectorize(seq)
        vecs[n][np.isin(kmerbasis, present)] = 1
        seqs.append(V.reduce(seq, alphabet=alphabet, mapping=A.FULL_ALPHABETS))
        ids.append(rid)
        lengths.append(len(seq))
    return dict(
        kmerlist=np.asarray(kmerbasis, dtype=str),
        ids=np.asarray(ids, dtype=str),
        seqs=np.asarray(seqs, dtype=str),
        vecs=vecs,
        lengths=np.asarray(lengths),
    )


def run_counts(seqs, kmerlist):
    """Count projection as rules/learn.smk:359-383 / rules/apply.smk:188-206 define it:
    all length-k substrings of the reduced string, looked up in kmerlist order."""
    k = len(kmerlist[0])
    rows = []
    for v in seqs:
        v = str(v)
        tally = Counter(v[i : i + k] for i in range(0, len(v) - k + 1))
        rows.append([tally.get(str(km), 0) for km in kmerlist])
    return np.asarray(rows, dtype=np.int64).reshape(len(seqs), len(kmerlist))


def to_csr(M):
    rows, cols = np.nonzero(M)
    rowptr = np.zeros(M.shape[0] + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32), M[rows, cols].astype(np.int32)


def main(ref_root):
    A, V, S, U = import_reference(ref_root)
    from sklearn.metrics.pairwise import cosine_similarity
    import scipy.sparse as sp

    A.ALPHABETS["red6"] = dict(RED6)
    A.FULL_ALPHABETS["red6"] = {}
    for grp, letter in RED6.items():
        if grp != "_keys":
            A.FULL_ALPHABETS["red6"].update({c: letter for c in grp})

    alpha_names = ["hydro", "standard", "solvacc", "hydrocharge", "hydrostruct", "miqs", "ptm", "None", "red6"]

    # ---- G1: alphabet tables -------------------------------------------------------
    g1 = {}
    for name in alpha_names:
        g1[name] = {
            "full": A.FULL_ALPHABETS[name],
            "char_set": sorted(A.get_alphabet_keys(name)),
            "short": A.get_alphabet(name),
        }
    g1["_meta"] = {
        "ALPHABET_ORDER": {str(k): v for k, v in A.ALPHABET_ORDER.items()},
        "ALPHABET2ID": A.ALPHABET2ID,
        "ALPHABET_ID": A.ALPHABET_ID,
        "StandardAlphabet": A.StandardAlphabet,
        "check_valid_error": None,
    }
    try:
        A.check_valid("no-such-alphabet")
    except ValueError as e:
        g1["_meta"]["check_valid_error"] = str(e)
    json.dump(g1, open(os.path.join(HERE, "g1_alphabets.json"), "w"), indent=1, sort_keys=True)

    # ---- G2: edge strings ----------------------------------------------------------
    g2 = []
    for name in alpha_names:
        for s in EDGE_STRINGS:
            red = V.reduce(s, alphabet=name, mapping=A.FULL_ALPHABETS)
            for k in EDGE_KS:
                out = V.KmerVec(name, k).reduce_vectorize(s)
                g2.append(
                    {
                        "alphabet": name,
                        "k": k,
                        "seq": s,
                        "reduced": red,
                        "kmers": [str(x) for x in out],
                        "dtype": str(out.dtype),
                        "shape": list(out.shape),
                    }
                )
    json.dump(g2, open(os.path.join(HERE, "g2_edge_cases.json"), "w"), ensure_ascii=True)

    # ---- demo FASTA inputs (data files the reference ships) -------------------------
    demo_dir = os.path.join(ref_root, "resources", "tutorial", "demo_example", "input")
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    demo_files = sorted(f for f in os.listdir(demo_dir) if f.endswith(".faa"))
    records, file_of = [], []
    for fi, f in enumerate(demo_files):
        shutil.copyfile(os.path.join(demo_dir, f), os.path.join(HERE, "data", f))
        os.chmod(os.path.join(HERE, "data", f), 0o644)
        recs = read_fasta(os.path.join(demo_dir, f))
        records += recs
        file_of += [fi] * len(recs)
    file_of = np.asarray(file_of)

    # ---- G3/G4/G5: rule outputs, counts, cosine -------------------------------------
    configs = [
        ("hydro", 14, 0),
        ("hydro", 14, 1),
        ("hydro", 14, 2),
        ("standard", 8, 0),
        ("red6", 8, 0),
        ("hydro", 20, 0),
        ("solvacc", 8, 0),
        ("miqs", 8, 0),
        ("hydrocharge", 6, 0),
        (None, 3, 0),
    ]
    summary = {}
    for alphabet, k, mf in configs:
        out = run_rule(V, A, records, alphabet, k, min_filter=mf)
        counts = run_counts(out["seqs"], out["kmerlist"])
        assert ((counts > 0) == (out["vecs"] > 0)).all()
        cos = cosine_similarity(counts, counts)
        totals = np.vstack([counts[file_of == fi].sum(axis=0) for fi in range(len(demo_files))])
        cos_rect = cosine_similarity(totals, counts).T
        rp, ci, cv = to_csr(counts)
        tag = f"{alphabet}_k{k}_mf{mf}"
        np.savez_compressed(
            os.path.join(HERE, f"g3_demo_{tag}.npz"),
            kmerlist=out["kmerlist"],
            ids=out["ids"],
            seqs=out["seqs"],
            lengths=out["lengths"],
            vecs_bits=np.packbits(out["vecs"].astype(bool), axis=1),
            vecs_shape=np.asarray(out["vecs"].shape),
            counts_rowptr=rp,
            counts_col=ci,
            counts_val=cv,
            cosine=cos,
            totals=totals,
            cosine_rect=cos_rect,
            file_of=file_of,
        )
        summary[tag] = dict(
            basis=int(len(out["kmerlist"])),
            nnz=int((counts > 0).sum()),
            total=int(counts.sum()),
            max=int(counts.max()) if counts.size else 0,
            first=str(out["kmerlist"][0]) if len(out["kmerlist"]) else None,
        )

    # basis.txt branch (kmerize.smk:72-78): explicit basis, file order, no filter
    ref0 = run_rule(V, A, records, "hydro", 14)
    explicit = [str(x) for x in ref0["kmerlist"][::7][:200]] + ["SSSSSSSSSSSSSS", "VVVVVVVVVVVVVX"]
    out = run_rule(V, A, records, "hydro", 14, basis=explicit)
    counts = run_counts(out["seqs"], out["kmerlist"])
    rp, ci, cv = to_csr(counts)
    np.savez_compressed(
        os.path.join(HERE, "g3_demo_hydro_k14_basisfile.npz"),
        kmerlist=out["kmerlist"],
        ids=out["ids"],
        seqs=out["seqs"],
        lengths=out["lengths"],
        vecs_bits=np.packbits(out["vecs"].astype(bool), axis=1),
        vecs_shape=np.asarray(out["vecs"].shape),
        counts_rowptr=rp,
        counts_col=ci,
        counts_val=cv,
    )
    # the same branch with k-mers listed more than once (kmerize.smk:72-78 takes the lines verbatim; np.isin at :119 sets
    # every column that carries a present k-mer, and the count loop of learn.smk:376-383 looks every column up by name)
    dup = explicit[:40] + [explicit[0], explicit[5], "SSSSSSSSSSSSSS", explicit[0]] + explicit[40:60] + [explicit[59]]
    out = run_rule(V, A, records, "hydro", 14, basis=dup)
    counts = run_counts(out["seqs"], out["kmerlist"])
    rp, ci, cv = to_csr(counts)
    np.savez_compressed(
        os.path.join(HERE, "g3_demo_hydro_k14_basisdup.npz"),
        kmerlist=out["kmerlist"],
        vecs_bits=np.packbits(out["vecs"].astype(bool), axis=1),
        vecs_shape=np.asarray(out["vecs"].shape),
        counts_rowptr=rp,
        counts_col=ci,
        counts_val=cv,
    )
    json.dump(summary, open(os.path.join(HERE, "g3_summary.json"), "w"), indent=1, sort_keys=True)

    # ---- G6: KmerBasis.transform ----------------------------------------------------
    rng = np.random.default_rng(7)
    kb = V.KmerBasis()
    basis = ["AAA", "AAC", "ACA", "CCC", "CAC", "ZZZ"]
    kb.set_basis(basis)
    vec_basis = ["CAC", "AAA", "QQQ", "ACA"]
    mat = rng.integers(0, 5, size=(4, len(vec_basis))).astype(float)
    g6 = {
        "basis": basis,
        "vector_basis": vec_basis,
        "matrix": mat.tolist(),
        "out": kb.transform(mat, vec_basis).tolist(),
        "errors": {},
    }
    for label, fn in {
        "set_basis_type": lambda: V.KmerBasis().set_basis(5),
        "vector_basis_type": lambda: kb.transform(mat, 5),
        "shape_mismatch": lambda: kb.transform(mat, vec_basis[:2]),
        "one_dim": lambda: kb.transform(mat[0], vec_basis),
    }.items():
        try:
            fn()
            g6["errors"][label] = None
        except Exception as e:  # noqa
            g6["errors"][label] = [type(e).__name__, str(e)]
    kv = V.KmerVec("hydro", 3)
    kv.set_kmer_set(["SSS", "SSV", "VVV"])
    g6["harmonize"] = kv.harmonize(np.array([[1.0, 2.0], [3.0, 4.0]]), ["VVV", "SVS"]).tolist()
    g6["kmervec_attrs"] = sorted(kv.__dict__.keys())
    g6["kmerset_attrs"] = sorted(kv.kmer_set.__dict__.keys())
    g6["kmerset_kmers"] = list(kv.kmer_set.kmers)
    g6["snekmer_version"] = kv.snekmer_version
    json.dump(g6, open(os.path.join(HERE, "g6_basis.json"), "w"), indent=1)

    # ---- G7: make_feature_matrix ----------------------------------------------------
    g7 = []
    kvs = V.KmerVec("standard", 3)
    ragged = [kvs.reduce_vectorize(s) for s in EDGE_STRINGS[3:12]]
    for mf in (0, 1, 2):
        res, kl = V.make_feature_matrix(ragged, min_filter=mf)
        g7.append(
            {
                "min_filter": mf,
                "vecs": [[str(x) for x in r] for r in ragged],
                "kmerlist": [str(x) for x in kl],
                "rows": [r.tolist() for r in res],
            }
        )
    json.dump(g7, open(os.path.join(HERE, "g7_feature_matrix.json"), "w"))

    # ---- G8: seeded synthetic families ------------------------------------------------
    from snekmer_amd.synth import synth_families, to_records

    for alphabet, k, idx in (("red6", 12, 2), ("standard", 12, 2), ("hydro", 20, 5)):
        res, off, fam = synth_families(256, 300, family=16, seed=20250523 + idx)
        recs = to_records(res, off)
        out = run_rule(V, A, recs, alphabet, k)
        counts = run_counts(out["seqs"], out["kmerlist"])
        cos = cosine_similarity(sp.csr_matrix(counts), sp.csr_matrix(counts))
        rp, ci, cv = to_csr(counts)
        np.savez_compressed(
            os.path.join(HERE, f"g8_synth_{alphabet}_k{k}.npz"),
            residues=res,
            offsets=off,
            family=fam,
            kmerlist=out["kmerlist"],
            counts_rowptr=rp,
            counts_col=ci,
            counts_val=cv,
            cosine=np.asarray(cos),
            lengths=out["lengths"],
        )

    # ---- G10: a real proteome from the reference's CI data (.test/input_learnapp, config k=8 alphabet=2)
    prot = os.path.join(ref_root, ".test", "input_learnapp", "UP000322080_2603819.fasta")
    shutil.copyfile(prot, os.path.join(HERE, "data", "UP000322080_2603819.fasta"))
    os.chmod(os.path.join(HERE, "data", "UP000322080_2603819.fasta"), 0o644)
    precs = read_fasta(prot)
    out = run_rule(V, A, precs, 2, 8)
    counts = run_counts(out["seqs"], out["kmerlist"])
    assert ((counts > 0) == (out["vecs"] > 0)).all()
    sample = np.sort(np.random.default_rng(10).choice(len(precs), size=24, replace=False))
    cos_rows = cosine_similarity(counts[sample], counts)
    np.savez_compressed(
        os.path.join(HERE, "g10_proteome_solvacc_k8.npz"),
        kmerlist=out["kmerlist"],
        ids=out["ids"],
        lengths=out["lengths"],
        reduced_lengths=np.asarray([len(x) for x in out["seqs"]]),
        row_count_sums=counts.sum(axis=1),
        row_presence_sums=out["vecs"].sum(axis=1).astype(np.int64),
        col_totals=counts.sum(axis=0),
        col_df=(counts > 0).sum(axis=0),
        nnz=np.asarray([(counts > 0).sum()]),
        max_count=np.asarray([counts.max()]),
        sample_rows=sample,
        cosine_rows=cos_rows,
        sample_counts=counts[sample[:4]],
    )

    # ---- G11: Jaccard distance as scripts/cluster_cluster.py:189-190 computes it without BSF
    from scipy.spatial.distance import pdist, squareform

    demo = run_rule(V, A, records, "hydro", 14)
    np.savez_compressed(
        os.path.join(HERE, "g11_jaccard_demo_hydro_k14.npz"),
        jaccard_distance=squareform(pdist(demo["vecs"], "jaccard")),
    )

    # ---- G9: score.connection_matrix_from_features / utils.to_feature_matrix ---------
    X = rng.integers(0, 4, size=(9, 23)).astype(float)
    X[3] = 0
    np.savez_compressed(
        os.path.join(HERE, "g9_connection.npz"),
        X=X,
        cosine=S.connection_matrix_from_features(X, metric="cosine"),
        jaccard=S.connection_matrix_from_features(X > 0, metric="jaccard"),
        tfm=U.to_feature_matrix([list(r) for r in X], length_array=np.arange(1, 10)),
        tfm_default=U.to_feature_matrix([list(r) for r in X]),
    )
    print(json.dumps(summary, indent=1))
    extra(ref_root)


def apply_rule(pd, cosine_similarity, totals_df, counts_df, conf_csv_path):
    """rules/apply.smk:278-328 around the real sklearn / pandas calls: cosine of the family totals
    against the query counts, top-2 by argsort, Score / delta / Prediction / Confidence."""
    cosine_df = cosine_similarity(totals_df, counts_df).T
    kct = pd.DataFrame(cosine_df, columns=totals_df.index, index=counts_df.index)
    gcs = pd.read_csv(str(conf_csv_path))
    gcs.index = gcs[gcs.columns[0]]
    gcs = gcs.iloc[:, 1:]
    gcs = gcs[gcs.columns[0]].squeeze()
    score_rank = []
    sorted_vals = np.argsort(-kct.values, axis=1)[:, :2]
    for i, item in enumerate(sorted_vals):
        # apply.smk:315-319 writes columns[[item]] (a 1 x 2 index, rejected by pandas >= 2): the two
        # columns `item` of row i
        score_rank.append((kct[kct.columns[item]][i : i + 1]).values.tolist()[0])
    delta = [score[0] - score[1] for score in score_rank]
    top_score = [score[0] for score in score_rank]
    vals = pd.DataFrame({"delta": delta})
    # apply.smk:318 indexes the column Index with the 2-D array, which pandas >= 2 rejects; older
    # pandas returned the 2-D label array that indexing the labels as an ndarray gives
    predictions = pd.DataFrame(np.asarray(kct.columns)[sorted_vals][:, :1])
    predictions.columns = ["Prediction"]
    predictions = predictions.astype(str)
    vals = vals.round(decimals=2)
    vals["Confidence"] = vals["delta"].map(gcs)
    return dict(
        sorted_vals=sorted_vals.astype(np.int64),
        score_rank=np.asarray(score_rank, dtype=np.float64),
        Score=np.asarray(top_score, dtype=np.float64),
        delta=vals["delta"].to_numpy(dtype=np.float64),
        Confidence=vals["Confidence"].to_numpy(dtype=np.float64),
        Prediction=predictions["Prediction"].to_numpy(dtype=str),
    )


def extra(ref_root):
    """G12 (apply epilogue incl. the confidence lookup) and G13 (real-valued feature matrices)."""
    import tempfile

    import pandas as pd
    from sklearn.metrics.pairwise import cosine_similarity

    A, V, S, U = import_reference(ref_root)
    if "red6" not in A.ALPHABETS:
        A.ALPHABETS["red6"] = dict(RED6)
        A.FULL_ALPHABETS["red6"] = {r: c for grp, c in RED6.items() if grp != "_keys" for r in grp}
    from snekmer_amd.synth import synth_families, to_records

    # the global confidence table in the format rules/learn.smk's evaluate step writes
    # (index "Difference" 0.00 .. 1.00, columns confidence / weight / sum)
    diffs = [round(x * 0.01, 2) for x in range(0, 101)]
    conf = pd.DataFrame(
        {"confidence": [round(0.35 + 0.65 * (1 - np.exp(-6 * d)), 6) for d in diffs], "weight": [5000] * 101,
         "sum": [int(400 * np.exp(-3 * d)) for d in diffs]},
        index=pd.Index(diffs, name="Difference"),
    )
    conf = conf.drop(index=[0.57, 0.93])  # holes: Series.map then yields NaN
    tmp = tempfile.mkdtemp()
    conf_path = os.path.join(tmp, "global-confidence-scores.csv")
    conf.to_csv(conf_path)
    conf_text = open(conf_path).read()

    for alphabet, k, idx, fam_size in (("standard", 12, 2, 16), ("hydro", 14, 7, 8), ("solvacc", 8, 8, 16)):
        res, off, fam = synth_families(256, 300, family=fam_size, seed=20250523 + idx)
        recs = to_records(res, off)
        out = run_rule(V, A, recs, alphabet, k)
        counts = run_counts(out["seqs"], out["kmerlist"])
        nfam = int(fam.max()) + 1
        # learn on the even-numbered sequences (family totals, learn.smk:385-408), apply to all of them
        train = np.arange(len(recs)) % 2 == 0
        totals = np.vstack([counts[train & (fam == f)].sum(axis=0) for f in range(nfam)])
        names = [f"FAM{f:03d}" for f in range(nfam)]
        totals_df = pd.DataFrame(totals, index=names, columns=list(out["kmerlist"]))
        counts_df = pd.DataFrame(counts, index=list(out["ids"]), columns=list(out["kmerlist"]))
        r = apply_rule(pd, cosine_similarity, totals_df, counts_df, conf_path)
        np.savez_compressed(
            os.path.join(HERE, f"g12_apply_{alphabet}_k{k}.npz"),
            residues=res, offsets=off, family=fam, train=train, names=np.asarray(names),
            kmerlist=out["kmerlist"], confidence_csv=np.asarray(conf_text), **r,
        )

    # ---- G13: real-valued features ------------------------------------------------------
    rng = np.random.default_rng(13)
    X = rng.normal(size=(37, 53))
    X[5] = 0.0
    Y = rng.normal(size=(11, 53)) * rng.uniform(0.1, 30.0, size=(11, 1))
    demo_dir = os.path.join(ref_root, "resources", "tutorial", "demo_example", "input")
    records = []
    for f in sorted(x for x in os.listdir(demo_dir) if x.endswith(".faa")):
        records += read_fasta(os.path.join(demo_dir, f))
    out = run_rule(V, A, records, "hydro", 14)
    counts = run_counts(out["seqs"], out["kmerlist"])
    F = U.to_feature_matrix([list(r) for r in counts], length_array=out["lengths"])  # utils.py:183-203
    np.savez_compressed(
        os.path.join(HERE, "g13_float_features.npz"),
        X=X, Y=Y,
        cos_xx=cosine_similarity(X, X), cos_xy=cosine_similarity(X, Y),
        conn_cosine_x=S.connection_matrix_from_features(X, metric="cosine"),
        demo_lengths=out["lengths"],
        demo_conn_cosine=S.connection_matrix_from_features(F, metric="cosine"),
        demo_cos=cosine_similarity(F, F),
    )
    # ---- G14: a .kmers file as the reference writes it (rules/kmerize.smk:141-142: pickle.dump(kmer, f))
    import pickle

    kmer = V.KmerVec(alphabet="hydro", k=14)
    kmer.set_kmer_set(out["kmerlist"][:200])
    blob = pickle.dumps(kmer, protocol=4)
    # the reverse direction: a .kmers stream written by this repository (snekmer_amd.io.dump_kmers, host code only) is
    # unpickled HERE by the imported reference, as snekmer/io.py:21-39 does, and read the way scripts/cluster_cluster.py:53-63
    # reads it; the outcome is recorded as data in the fixture
    import io as _io

    import snekmer_amd

    mine = snekmer_amd.vectorize.KmerVec("hydro", 14)
    mine.set_kmer_set(out["kmerlist"][:200])
    buf = _io.BytesIO()
    snekmer_amd.io.dump_kmers(mine, buf)
    back = pickle.loads(buf.getvalue())
    assert type(back) is V.KmerVec and type(back.basis) is V.KmerBasis and type(back.kmer_set) is V.KmerSet
    state_equal = (back.alphabet == kmer.alphabet and back.k == kmer.k and set(back.char_set) == set(kmer.char_set)
                   and back.vector is None and sorted(back.__dict__) == sorted(kmer.__dict__)
                   and (np.asarray(back.kmer_set._kmerlist) == np.asarray(kmer.kmer_set._kmerlist)).all()
                   and sorted(back.basis.__dict__) == sorted(kmer.basis.__dict__)
                   and all(np.array_equal(np.asarray(getattr(back.basis, a), dtype=object), np.asarray(getattr(kmer.basis, a), dtype=object))
                           for a in kmer.basis.__dict__))
    assert state_equal
    # and the reference's own methods work on the object it loaded (a 2 x 3 array over 3 of the k-mers moved onto the 200-k-mer basis)
    sub = [str(x) for x in out["kmerlist"][[5, 0, 150]]]
    moved = back.basis.transform(np.arange(1, 7).reshape(2, 3), sub)
    reverse = {"loaded_class": f"{type(back).__module__}.{type(back).__name__}", "state_equal": bool(state_equal),
               "stream_bytes": len(buf.getvalue()), "reference_stream_bytes": len(blob),
               "transform_nonzero_columns": [int(c) for c in np.flatnonzero(np.asarray(moved).any(axis=0))],
               "transform_values": np.asarray(moved)[:, np.flatnonzero(np.asarray(moved).any(axis=0))].tolist(), "transform_kmers": sub}
    json.dump({"pickle_hex": blob.hex(), "alphabet": "hydro", "k": 14, "char_set": sorted(kmer.char_set),
               "n_kmers": 200, "first": [str(x) for x in out["kmerlist"][:5]],
               "snekmer_version": kmer.snekmer_version, "attrs": sorted(kmer.__dict__.keys()), "reverse_direction": reverse},
              open(os.path.join(HERE, "g14_reference_kmers_pickle.json"), "w"))
    print("extra fixtures written: g12_apply_*.npz, g13_float_features.npz, g14_reference_kmers_pickle.json")


def hamming_counts(ref_root):
    """G15: the DEFAULT call of score.connection_matrix_from_features (metric="jaccard" -> 1 - hamming distance,
    snekmer/score.py:166-168) on the input its docstring names, a k-mer COUNT matrix, and on a real-valued one; plus
    scipy's Jaccard distance (scripts/cluster_cluster.py:189-190) on the same non-binary matrices."""
    from scipy.spatial.distance import pdist, squareform

    A, V, S, U = import_reference(ref_root)
    g9 = np.load(os.path.join(HERE, "g9_connection.npz"))
    X = g9["X"]  # counts 0..3, one all-zero row
    demo_dir = os.path.join(ref_root, "resources", "tutorial", "demo_example", "input")
    records = []
    for f in sorted(x for x in os.listdir(demo_dir) if x.endswith(".faa")):
        records += read_fasta(os.path.join(demo_dir, f))
    out = run_rule(V, A, records, "hydro", 14)
    counts = run_counts(out["seqs"], out["kmerlist"])  # 52 x 4941, counts up to 13
    rng = np.random.default_rng(15)
    # real-valued: few distinct values per column so that equal non-zero cells exist; a zero row; negative values
    F = rng.choice(np.asarray([0.0, 0.0, 0.0, 0.25, -1.5, 2.0, 1e-3]), size=(31, 47))
    F[7] = 0.0
    F[9] = F[8]
    np.savez_compressed(
        os.path.join(HERE, "g15_hamming_counts.npz"),
        F=F,
        default_x=S.connection_matrix_from_features(X),
        default_demo_counts=S.connection_matrix_from_features(counts),
        default_float=S.connection_matrix_from_features(F),
        default_lengthnorm=S.connection_matrix_from_features(U.to_feature_matrix([list(r) for r in counts], length_array=out["lengths"])),
        demo_lengths=out["lengths"],
        jaccard_x=squareform(pdist(X, "jaccard")),
        jaccard_demo_counts=squareform(pdist(counts, "jaccard")),
        jaccard_float=squareform(pdist(F, "jaccard")),
        demo_counts_checksum=np.asarray([counts.sum(), (counts > 0).sum(), counts.max()]),
    )
    print("g15_hamming_counts.npz written")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = args[0] if args else "/root/reference"
    if "--extra-only" in sys.argv:
        extra(root)
    elif "--g15-only" in sys.argv:
        hamming_counts(root)
    else:
        main(root)
        hamming_counts(root)
