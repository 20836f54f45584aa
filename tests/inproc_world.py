"""Test-only: G ranks of dist.ShardedPipeline as G threads of ONE process sharing one GPU.

RCCL refuses two ranks on one device (tools/probe_rccl_same_gpu.py), and a one-GPU box is what the
`-m gpu` suite gets.  Here every rank is a host thread with its own context (own HIP stream, own
scratch) running the unmodified ShardedPipeline; `ThreadExchange` has dist.RcclExchange's interface
and moves the bytes with device-to-device copies that EXECUTE THE C LIBRARY'S OWN PLANS
(skm_plan_alltoallv / skm_plan_allgatherv): rank r copies, for every (peer, array), the range
[send_off, send_off + send_bytes) the PEER's plan assigns to r into the range [recv_off, recv_off +
recv_bytes) its own plan expects from that peer, and asserts that the two sizes agree (the pairing
RCCL's grouped send/recv relies on).  Only the transport differs from the 8-GPU run; every kernel,
buffer size, hash table and offset is the one a real rank computes, at full per-rank size.
"""
import ctypes as C
import os
import threading
import time
import traceback

import numpy as np

from snekmer_amd import _hip

BARRIER_TIMEOUT_S = 600.0
_T0 = time.perf_counter()
_LOG_LOCK = threading.Lock()


def progress(msg: str) -> None:
    """Timestamped line in gpurun_out/inproc_world.log (when that directory exists: the GPU box) — the long full-size
    tests say where they are, and the runner's silence watchdog sees a file growing."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(root):
        return
    with _LOG_LOCK:
        with open(os.path.join(root, "inproc_world.log"), "a") as fh:
            fh.write(f"[+{time.perf_counter() - _T0:7.1f}s {threading.current_thread().name}] {msg}\n")


class ThreadWorld:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.errors = []
        self.bytes_moved = 0  # off-rank bytes, summed over ranks (what xGMI would carry)
        self._lock = threading.Lock()

    def wait(self):
        self.barrier.wait(timeout=BARRIER_TIMEOUT_S)

    def account(self, nbytes: int):
        with self._lock:
            self.bytes_moved += int(nbytes)


class ThreadExchange:
    """dist.RcclExchange's interface over a ThreadWorld."""

    def __init__(self, ctx, tw: ThreadWorld, rank: int):
        self.ctx, self.tw, self.world, self.rank = ctx, tw, tw.world, rank

    def allgather_i64(self, values) -> np.ndarray:
        vals = np.atleast_1d(np.asarray(values, dtype=np.int64)).copy()
        self.tw.slots[self.rank] = vals
        self.tw.wait()
        out = np.stack(self.tw.slots)
        self.tw.wait()
        return out

    def _run(self, ops, sends, recvs):
        na = len(sends)
        progress(f"exchange of {na} arrays: waiting for my stream")
        self.ctx.sync()  # this rank's send buffers are complete before a peer reads them
        self.tw.slots[self.rank] = (ops, [s.ptr for s in sends], [s.nbytes for s in sends])
        self.tw.wait()
        for p in range(self.world):
            peer_ops, peer_ptrs, peer_sizes = self.tw.slots[p]
            for a in range(na):
                mine, theirs = ops[p * na + a], peer_ops[self.rank * na + a]
                assert (mine.peer, mine.array, theirs.peer, theirs.array) == (p, a, self.rank, a)
                assert theirs.send_bytes == mine.recv_bytes, (self.rank, p, a, theirs.send_bytes, mine.recv_bytes)
                assert 0 <= theirs.send_off and theirs.send_off + theirs.send_bytes <= peer_sizes[a]
                assert 0 <= mine.recv_off and mine.recv_off + mine.recv_bytes <= recvs[a].nbytes
                if mine.recv_bytes:
                    self.ctx.call("skm_memcpy_d2d", C.c_void_p(recvs[a].ptr + mine.recv_off),
                                  C.c_void_p(peer_ptrs[a] + theirs.send_off), C.c_size_t(mine.recv_bytes))
                    if p != self.rank:
                        self.tw.account(mine.recv_bytes)
        self.ctx.sync()
        progress("exchange: my copies are done")
        self.tw.wait()  # nobody reuses a send buffer while a peer may still be reading it

    def alltoallv_multi(self, sends, recvs, elem_bytes, send_counts, recv_counts):
        na = len(sends)
        ops = (_hip.P2POp * (self.world * na))()
        eb, sc, rc = (np.ascontiguousarray(x, dtype=np.int64) for x in (elem_bytes, send_counts, recv_counts))
        p = C.c_void_p
        _hip._check(self.ctx.lib, self.ctx.lib.skm_plan_alltoallv(self.world, na, eb.ctypes.data_as(p), sc.ctypes.data_as(p),
                                                                 rc.ctypes.data_as(p), ops))
        self._run(ops, sends, recvs)

    def allgatherv_multi(self, sends, recvs, elem_bytes, counts):
        na = len(sends)
        ops = (_hip.P2POp * (self.world * na))()
        eb, cn = np.ascontiguousarray(elem_bytes, dtype=np.int64), np.ascontiguousarray(counts, dtype=np.int64)
        p = C.c_void_p
        _hip._check(self.ctx.lib, self.ctx.lib.skm_plan_allgatherv(self.world, self.rank, na, eb.ctypes.data_as(p),
                                                                  cn.ctypes.data_as(p), ops))
        self._run(ops, sends, recvs)

    # single-array forms (the replicated exchange)
    def allgatherv(self, d_send, nbytes_per_rank, d_recv):
        sizes = np.asarray(nbytes_per_rank, dtype=np.int64)
        self.allgatherv_multi([d_send], [d_recv], [1], [sizes])

    def alltoallv(self, d_send, send_bytes, d_recv, recv_bytes):
        self.alltoallv_multi([d_send], [d_recv], [1], send_bytes, recv_bytes)


def run_world(world: int, body, device: int = 0):
    """Run body(rank, ctx, exchange) on `world` threads; returns the list of results in rank order.  An exception on
    one rank aborts the barrier so that the others fail too instead of waiting."""
    tw = ThreadWorld(world)
    results = [None] * world

    def main(rank):
        try:
            ctx = _hip.Context(device)
            results[rank] = body(rank, ctx, ThreadExchange(ctx, tw, rank))
            ctx.sync()
        except BaseException as exc:  # noqa: BLE001
            tw.errors.append((rank, exc, traceback.format_exc()))
            tw.barrier.abort()

    threads = [threading.Thread(target=main, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in tw.errors if not isinstance(e[1], threading.BrokenBarrierError)] or tw.errors
    if real:
        rank, exc, tb = real[0]
        raise AssertionError(f"rank {rank} failed:\n{tb}") from exc
    return results, tw
