"""World-size-2 checks (gloo, CPU) of the host-side sharding plan in snekmer_amd/dist.py.

Two ranks each count the k-mers of their row block (with the oracle standing in for the device
kernel, which is what tests may do), exchange the CSR shards through gloo exactly as
ShardedPipeline does through RCCL (sizes first, then three variable-size all-gathers), rebuild
the row pointers with the host statement of skm_csr_concat_rowptr and compute their row block
of the cosine matrix.  The assembled result must equal the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import c_oracle
    from snekmer_amd import alphabet as A
    from snekmer_amd.dist import concat_rowptr_host, plan_allgather, shard_bounds
    from snekmer_amd.synth import synth_families

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if "red6" not in A.ALPHABETS:
            A.register_alphabet("red6", A.RED6_GROUPS)
        lut = A.build_lut("red6")
        k, n = 12, 301  # odd on purpose: uneven shards
        res, off, _ = synth_families(n, 300, family=25, seed=99)
        bounds = shard_bounds(n, world)
        lo, hi = bounds[rank]
        loc_rp, loc_codes, loc_counts, _ = c_oracle.count_csr(
            lut.rank, lut.nsym, k, res[off[lo] : off[hi]], off[lo : hi + 1] - off[lo]
        )

        def allgather_i64(v):
            t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(t, torch.tensor([v], dtype=torch.int64))
            return [int(x.item()) for x in t]

        def allgatherv(buf: np.ndarray, nbytes):
            raw = np.frombuffer(buf.tobytes(), dtype=np.uint8)
            assert raw.size == nbytes[rank]
            mx = max(nbytes)
            pad = np.zeros(mx, dtype=np.uint8)
            pad[: raw.size] = raw
            outs = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, torch.from_numpy(pad))
            return np.concatenate([o.numpy()[: nbytes[r]] for r, o in enumerate(outs)])

        nnz = allgather_i64(len(loc_codes))
        maxc = allgather_i64(int(loc_counts.max()) if len(loc_counts) else 0)
        rows = [b - a for a, b in bounds]
        narrow = max(maxc) <= 255  # counts travel as bytes, as ShardedPipeline does
        plan = plan_allgather(nnz, rows, 4, 1 if narrow else 4)
        codes = allgatherv(loc_codes.astype(np.uint32), plan["codes"]).view(np.uint32)
        cdt = np.uint8 if narrow else np.uint32
        counts = allgatherv(loc_counts.astype(cdt), plan["counts"]).view(cdt).astype(np.uint32)
        rp_all = allgatherv(loc_rp.astype(np.int64), plan["rowptr"]).view(np.int64)
        pieces, pos = [], 0
        for r in rows:
            pieces.append(rp_all[pos : pos + r + 1])
            pos += r + 1
        rowptr = concat_rowptr_host(pieces)

        # every rank now holds the full CSR; compare with the unsharded oracle
        f_rp, f_codes, f_counts, f_first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
        assert (rowptr == f_rp).all() and (codes == f_codes.astype(np.uint32)).all() and (counts == f_counts).all()
        b, _, _, _, col = c_oracle.basis(rowptr, codes.astype(np.uint64), counts, np.zeros(len(codes), np.uint32))
        block = c_oracle.cosine_rows(rowptr, col, counts, len(b), np.arange(lo, hi))
        np.save(os.path.join(outdir, f"block{rank}.npy"), block)
        dist.barrier()
        if rank == 0:
            full = np.concatenate([np.load(os.path.join(outdir, f"block{r}.npy")) for r in range(world)])
            fb, _, _, _, fcol = c_oracle.basis(f_rp, f_codes, f_counts, f_first)
            ref = c_oracle.cosine_rows(f_rp, fcol, f_counts, len(fb), np.arange(n))
            assert full.shape == ref.shape and np.abs(full - ref).max() < 1e-12
            open(os.path.join(outdir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _worker_owner(rank, world, port, outdir):
    """The distributed-basis exchange (ShardedPipeline(basis="distributed")) restated on the host:
    entries grouped by owner, all-to-all sized by plan_alltoall, owner postings (postings_host),
    all-gather, local column lookup, row block of the cosine matrix from the postings."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import c_oracle
    from snekmer_amd import alphabet as A
    from snekmer_amd.dist import owner_answers_host, owner_host, plan_alltoall, postings_host, shard_bounds
    from snekmer_amd.synth import synth_families

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if "red6" not in A.ALPHABETS:
            A.register_alphabet("red6", A.RED6_GROUPS)
        lut = A.build_lut("red6")
        k, n = 12, 211
        res, off, _ = synth_families(n, 300, family=25, seed=98)
        bounds = shard_bounds(n, world)
        lo, hi = bounds[rank]
        rp, codes, counts, _ = c_oracle.count_csr(lut.rank, lut.nsym, k, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
        codes = codes.astype(np.uint32)
        rows = np.repeat(np.arange(lo, hi, dtype=np.uint64), np.diff(rp))
        rowcount = rows | (counts.astype(np.uint64) << np.uint64(32))

        def allgather_obj(x):
            out = [None] * world
            dist.all_gather_object(out, x)
            return out

        def alltoallv(buf: np.ndarray, send_bytes, recv_bytes):
            raw = np.frombuffer(buf.tobytes(), dtype=np.uint8)
            so = np.concatenate([[0], np.cumsum(send_bytes)])
            outs = [torch.zeros(int(b), dtype=torch.uint8) for b in recv_bytes]
            reqs = []
            for p in range(world):
                seg = torch.from_numpy(raw[so[p]:so[p + 1]].copy())
                if p == rank:
                    outs[p].copy_(seg)
                    continue
                if seg.numel():
                    reqs.append(dist.isend(seg, dst=p))
                if outs[p].numel():
                    reqs.append(dist.irecv(outs[p], src=p))
            for r in reqs:
                r.wait()
            return np.concatenate([o.numpy() for o in outs]) if sum(recv_bytes) else np.zeros(0, np.uint8)

        # 1. group by owner (stable), 2. all-to-all
        own = owner_host(codes, world)
        order = np.argsort(own, kind="stable")
        cmat = np.asarray(allgather_obj(np.bincount(own, minlength=world)))
        sb, rb = plan_alltoall(cmat, rank, 4)
        r_codes = alltoallv(codes[order], sb, rb).view(np.uint32)
        sb, rb = plan_alltoall(cmat, rank, 8)
        r_rc = alltoallv(rowcount[order], sb, rb).view(np.uint64)
        assert len(r_codes) == cmat[:, rank].sum() and (owner_host(r_codes, world) == rank).all()
        # 3. owner postings, 4. all-gather
        distinct, o_code, o_start, o_post = postings_host(r_codes, r_rc)
        parts = allgather_obj((distinct, o_code, o_start, o_post))
        ncols_total = sum(p[0] for p in parts)
        post = np.concatenate([p[3] for p in parts])
        base = np.cumsum([0] + [len(p[3]) for p in parts])
        colptr = np.concatenate([p[2].astype(np.int64) + base[i] for i, p in enumerate(parts)] + [[base[-1]]])
        table = {}
        for p in parts:
            for c in p[1]:
                table[int(c)] = len(table)
        # the column ids as ShardedPipeline learns them (round 5): every owner answers the entries it received, one uint32
        # each in the order received, through the REVERSE of the all-to-all; the sender adds the owner's first global column
        sb, rb = plan_alltoall(cmat.T, rank, 4)
        back = alltoallv(owner_answers_host(r_codes), sb, rb).view(np.uint32)
        assert len(back) == len(codes)
        colbase = np.cumsum([0] + [len(p[1]) for p in parts])
        grp_owner = np.repeat(np.arange(world), cmat[rank, :])  # owner of every grouped position
        colidx = np.full(len(codes), 0xFFFFFFFF, dtype=np.int64)
        colidx[order] = np.where(back == 0xFFFFFFFF, 0xFFFFFFFF, back.astype(np.int64) + colbase[grp_owner])
        want = np.asarray([table.get(int(c), 0xFFFFFFFF) for c in codes], dtype=np.int64)
        assert (colidx == want).all()
        # 5. own rows against the postings; norms all-gathered
        norms = np.concatenate(allgather_obj(np.sqrt(np.add.reduceat(counts.astype(np.float64) ** 2, rp[:-1]))
                                             if len(counts) else np.zeros(hi - lo)))
        inv = np.where(norms > 0, 1.0 / np.where(norms > 0, norms, 1.0), 1.0)
        block = np.zeros((hi - lo, n))
        for i in range(lo, hi):
            for e in range(rp[i - lo], rp[i - lo + 1]):
                c = table.get(int(codes[e]))
                v = float(counts[e])
                if c is None:
                    block[i - lo, i] += v * v
                else:
                    seg = post[colptr[c]:colptr[c + 1]]
                    js = (seg & np.uint64(0xFFFFFFFF)).astype(np.int64)
                    assert (np.diff(js) > 0).all()  # rows ascending within a column
                    block[i - lo, js] += v * (seg >> np.uint64(32)).astype(np.float64)
        block *= inv[lo:hi, None] * inv[None, :]
        np.save(os.path.join(outdir, f"oblock{rank}.npy"), block)
        dist.barrier()
        if rank == 0:
            full = np.concatenate([np.load(os.path.join(outdir, f"oblock{r}.npy")) for r in range(world)])
            f_rp, f_codes, f_counts, f_first = c_oracle.count_csr(lut.rank, lut.nsym, k, res, off)
            fb, _, _, _, fcol = c_oracle.basis(f_rp, f_codes, f_counts, f_first)
            assert ncols_total == len(fb)
            ref = c_oracle.cosine_rows(f_rp, fcol, f_counts, len(fb), np.arange(n))
            assert full.shape == ref.shape and np.abs(full - ref).max() < 1e-12
            open(os.path.join(outdir, "ok_owner"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_owner_exchange_plan(tmp_path, world):
    import torch.multiprocessing as mp

    mp.spawn(_worker_owner, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok_owner").exists()


def test_plan_alltoall_and_owner_hash():
    from snekmer_amd.dist import owner_host, plan_alltoall

    cmat = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]])
    sb, rb = plan_alltoall(cmat, 1, 4)
    assert sb.tolist() == [16, 20, 24] and rb.tolist() == [8, 20, 32]
    rng = np.random.default_rng(3)
    for dt, hi in ((np.uint32, 6**12), (np.uint64, 7**12)):
        codes = rng.integers(0, hi, size=200000).astype(dt)
        for w in (1, 2, 3, 8):
            o = owner_host(codes, w)
            assert o.min() >= 0 and o.max() < w
            assert np.bincount(o, minlength=w).min() > 0.9 * len(codes) / w  # balanced
    # neighbouring codes (low-complexity k-mers) spread over owners
    assert len(set(owner_host(np.arange(64, dtype=np.uint32), 8).tolist())) == 8


def test_two_rank_csr_exchange_plan(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_bounds_cover_and_balance():
    from snekmer_amd.dist import concat_rowptr_host, plan_allgather, shard_bounds, shard_bounds_by_residues

    for n, w in [(0, 2), (1, 2), (7, 8), (100000, 8), (100001, 3)]:
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
    off = np.concatenate([[0], np.cumsum(np.r_[np.full(50, 1000), np.full(950, 100)])])
    b = shard_bounds_by_residues(off, 4)
    assert b[0][0] == 0 and b[-1][1] == 1000 and all(b[i][1] == b[i + 1][0] for i in range(3))
    res = [off[hi] - off[lo] for lo, hi in b]
    assert max(res) - min(res) <= 2 * 1000  # within two of the longest sequences
    plan = plan_allgather([10, 0, 5], [3, 2, 4], 8)
    assert plan == {"codes": [80, 0, 40], "counts": [40, 0, 20], "rowptr": [32, 24, 40]}
    assert plan_allgather([10, 0, 5], [3, 2, 4], 4, 1)["counts"] == [10, 0, 5]
    rp = concat_rowptr_host([np.array([0, 2, 5]), np.array([0]), np.array([0, 1, 1, 4])])
    assert rp.tolist() == [0, 2, 5, 6, 6, 9]


def test_c_plans_of_the_grouped_collectives_match_the_host_statements():
    """skm_plan_alltoallv / skm_plan_allgatherv (pure host functions of the C library: the byte ranges RCCL is
    handed) against dist.plan_*_host for world sizes 1..8, random segment tables, several arrays; and the
    pairing property RCCL relies on: what rank a sends to b is exactly what b expects from a."""
    import ctypes as C

    from snekmer_amd import _hip, dist as D

    lib = _hip.load_library()
    rng = np.random.default_rng(8)
    for world in range(1, 9):
        for trial in range(6):
            na = int(rng.integers(1, 6))
            eb = rng.choice([1, 2, 4, 8], size=na).astype(np.int64)
            cmat = rng.integers(0, 50, size=(world, world)).astype(np.int64)  # [src, dst] element counts
            if trial == 0:
                cmat[:] = 0
            plans = []
            for me in range(world):
                ops = (_hip.P2POp * (world * na))()
                sc, rc = np.ascontiguousarray(cmat[me, :]), np.ascontiguousarray(cmat[:, me])
                assert lib.skm_plan_alltoallv(world, na, eb.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                              rc.ctypes.data_as(C.c_void_p), ops) == 0
                ref = D.plan_alltoallv_host(eb, sc, rc)
                for p in range(world):
                    for a in range(na):
                        o = ops[p * na + a]
                        assert (o.peer, o.array) == (p, a)
                        assert (o.send_off, o.send_bytes, o.recv_off, o.recv_bytes) == ref[p][a]
                # single-array form agrees with the byte plan the first round used
                sb, rb = D.plan_alltoall(cmat, me, int(eb[0]))
                assert [ops[p * na].send_bytes for p in range(world)] == sb.tolist()
                assert [ops[p * na].recv_bytes for p in range(world)] == rb.tolist()
                plans.append(ops)
            for a_ in range(world):
                for b_ in range(world):
                    for a in range(na):
                        assert plans[a_][b_ * na + a].send_bytes == plans[b_][a_ * na + a].recv_bytes
            # all-gather
            counts = rng.integers(0, 40, size=(na, world)).astype(np.int64)
            for me in range(world):
                ops = (_hip.P2POp * (world * na))()
                assert lib.skm_plan_allgatherv(world, me, na, eb.ctypes.data_as(C.c_void_p),
                                               np.ascontiguousarray(counts).ctypes.data_as(C.c_void_p), ops) == 0
                ref = D.plan_allgatherv_host(me, eb, counts)
                for p in range(world):
                    for a in range(na):
                        o = ops[p * na + a]
                        assert (o.send_off, o.send_bytes, o.recv_off, o.recv_bytes) == ref[p][a]
                        assert o.recv_bytes == counts[a, p] * eb[a]
    bad = (_hip.P2POp * 4)()
    z = np.zeros(2, np.int64)
    assert lib.skm_plan_alltoallv(2, 1, np.asarray([0], np.int64).ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                                  z.ctypes.data_as(C.c_void_p), bad) == -1


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment must start its own ranks (a fresh
    torch.distributed.run child, before anything touches a GPU) and hand back the job's exit code.  Without a GPU the
    ranks fail loudly (HipUnavailable: no CPU fallback); the point here is that the failure comes from the ranks and
    not from argument handling."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box this test stays a CPU test
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "self-launch" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "HipUnavailable" in r.stderr and "rank" in r.stderr.lower()
    assert '"metric"' not in r.stdout
