"""GPU: the library's array pool, stream cache and asynchronous uploads (snekmer_amd/csrc/skm_mem.hip).

Round 5 ended with three long fuzz runs that stopped inside hipFree: skm_free no longer calls it, nothing
relies on its implicit device-wide wait, and contexts no longer create and destroy streams.  These tests pin that."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ensure_red6

pytestmark = pytest.mark.gpu

ensure_red6()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from snekmer_amd import _hip

    return _hip.default_context()


def test_score_calls_reuse_parked_blocks_and_never_call_hipfree(ctx):
    """The sklearn call sites (snekmer/score.py:149-172) allocate a dozen arrays per call: after a warm-up call the same
    calls are served from parked blocks - no hipMalloc, no hipFree - and give the same result."""
    import snekmer_amd as skm

    rng = np.random.default_rng(5)
    X = (rng.random((150, 1024)) < 0.01) * rng.integers(1, 5, size=(150, 1024))
    first = skm.score.connection_matrix_from_features(X)
    ctx.sync()
    before = ctx.mem_stats()
    for _ in range(20):
        again = skm.score.connection_matrix_from_features(X)
        assert (again == first).all()
    after = ctx.mem_stats()
    assert after["hipFree_calls"] == before["hipFree_calls"]
    assert after["hipMalloc_calls"] == before["hipMalloc_calls"], (before, after)
    assert after["reused"] > before["reused"]


def test_array_freed_while_its_reader_is_queued_is_not_handed_out_again(ctx):
    """skm_free while a kernel that reads the array is still queued: the block is parked behind an event, a request of the
    same size made at once gets OTHER memory, and the kernel's result is the one computed from the original contents."""
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.synth import synth_families

    lut = A.build_lut("red6")
    n = 20000
    res, off, _ = synth_families(n, 300, family=50, seed=11)
    pipe = engine.Pipeline(ctx, lut, 12)
    out = pipe.step(engine.SeqBatch(ctx, res, off))  # ~1.6 GB of output: the stream stays busy for a while
    want_first = out.download(64).copy()
    probe = ctx.to_device(np.arange(1 << 20, dtype=np.uint32))
    ctx.sync()
    pipe.cosine()  # queued, not waited for
    busy_ptr = probe.ptr
    cp = ctx.empty(1 << 20, np.uint32)
    ctx.call("skm_memcpy_d2d", __import__("ctypes").c_void_p(cp.ptr), __import__("ctypes").c_void_p(probe.ptr), __import__("ctypes").c_size_t(4 << 20))
    probe.free()  # its reader (the copy above) is queued behind the cosine
    other = ctx.empty(1 << 20, np.uint32)
    assert other.ptr != busy_ptr
    ctx.call("skm_memset", __import__("ctypes").c_void_p(other.ptr), 0xFF, __import__("ctypes").c_size_t(4 << 20))
    assert (cp.download() == np.arange(1 << 20, dtype=np.uint32)).all()
    assert (pipe.out.download(64) == want_first).all()
    ctx.sync()
    again = ctx.empty(1 << 20, np.uint32)  # now idle: a parked block of the class comes back
    stats = ctx.mem_stats()
    assert stats["reused"] >= 1 and again.ptr is not None


def test_contexts_recycle_their_streams(ctx):
    """engine.OverlappedPipeline opens two CU-confined side contexts per pipeline; closing and reopening them creates no
    new streams after the first time (the stop of round 5's fuzz followed hundreds of created and destroyed masked streams)."""
    from snekmer_amd import _hip

    a = _hip.Context(ctx.device, cu_groups=(0, 3))
    b = _hip.Context(ctx.device, cu_groups=(0, 3))
    a.close()
    b.close()
    cached = ctx.mem_stats()["cached_streams"]
    assert cached >= 2
    for _ in range(10):
        a = _hip.Context(ctx.device, cu_groups=(0, 3))
        b = _hip.Context(ctx.device, cu_groups=(0, 3))
        assert ctx.mem_stats()["cached_streams"] == cached - 2
        x = a.to_device(np.arange(100, dtype=np.int64))
        assert (x.download() == np.arange(100)).all()
        a.close()
        b.close()
        x.free()  # an array that outlives its context goes back through the default one
    assert ctx.mem_stats()["cached_streams"] == cached


def test_trim_returns_parked_memory(ctx):
    a = ctx.empty(64 << 20, np.uint8)
    a.free()
    ctx.sync()
    before = ctx.mem_stats()
    released = ctx.trim()
    after = ctx.mem_stats()
    assert released >= 64 << 20 and after["parked_bytes"] == 0 and after["hipFree_calls"] > before["hipFree_calls"]


def test_guard_mode_reports_an_overrun():
    """SKM_GUARD=1: 512 canary bytes behind every array; a write past the end is reported when the array is freed."""
    code = (
        "import ctypes as C, numpy as np\n"
        "from snekmer_amd import _hip\n"
        "ctx = _hip.default_context()\n"
        "ok = ctx.empty(1000, np.uint8)\n"
        "ctx.call('skm_memset', C.c_void_p(ok.ptr), 1, C.c_size_t(1000))\n"
        "ok.free()\n"
        "assert not _hip.FREE_ERRORS, _hip.FREE_ERRORS\n"
        "bad = ctx.empty(1000, np.uint8)\n"
        "ctx.call('skm_memset', C.c_void_p(bad.ptr), 1, C.c_size_t(1003))\n"
        "bad.free()\n"
        "assert len(_hip.FREE_ERRORS) == 1 and '1000 bytes' in _hip.FREE_ERRORS[0] and 'byte 0 ' in _hip.FREE_ERRORS[0], _hip.FREE_ERRORS\n"
        "print('guard ok')\n"
    )
    env = dict(os.environ, SKM_GUARD="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "guard ok" in r.stdout, r.stdout + r.stderr


def test_debug_report_names_contexts_and_pool(ctx):
    from snekmer_amd import _hip

    text = _hip.debug_report()
    assert "device 0" in text and "stream idle" in text or "stream BUSY" in text, text


def test_batch_uploader_stream_equals_resident_batches(ctx):
    """A stream of DIFFERENT batches uploaded asynchronously from pinned host memory into recycled device buffers
    (engine.BatchUploader, three slots, five batches: every slot is refilled) through engine.OverlappedPipeline: every
    step's matrix equals the one-stream Pipeline's on a resident copy of the same batch, bit for bit.  The reference reads
    every batch from a file (rules/kmerize.smk:89-129)."""
    from snekmer_amd import alphabet as A
    from snekmer_amd import engine
    from snekmer_amd.synth import synth_families

    lut = A.build_lut("red6")
    k = 12
    sizes = (3000, 2500, 3000, 1800, 2900)
    packed = [synth_families(n, 300, family=25, seed=100 + i)[:2] for i, n in enumerate(sizes)]
    single = engine.Pipeline(ctx, lut, k, dense_route=False)
    want = []
    for res, off in packed:
        out = single.step(engine.SeqBatch(ctx, res, off))
        n = len(off) - 1
        want.append(out.download().reshape(out.shape)[:n, :n].copy())
    up = engine.BatchUploader(ctx, max(int(r.size) for r, _ in packed), max(sizes), slots=3)
    pipe = engine.OverlappedPipeline(ctx, lut, k, side_list_fraction=0.6)
    pipe.SPLIT_MIN_ROWS = 1
    pipe.prefetch(up.upload(*packed[0]))
    for i in range(len(packed)):
        nxt = up.upload(*packed[i + 1]) if i + 1 < len(packed) else None
        out = pipe.step(nxt)
        n = sizes[i]
        got = out.download().reshape(out.shape)[:n, :n]  # (waits for the main stream only: the next upload may still run)
        assert (got == want[i]).all(), i
    pipe.sync()
    assert up.host_waits <= len(packed)
    pipe.close()
    up.close()
