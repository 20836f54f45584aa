#!/usr/bin/env python3
"""Probe: can two processes open RCCL communicators on the SAME GPU (one-GPU boxes cannot test nranks > 1 otherwise)?
Prints what skm_comm_init answers on each rank."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, world, port):
    import numpy as np
    import torch.distributed as dist

    from snekmer_amd import _hip
    from snekmer_amd.dist import RcclExchange

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    uid = [RcclExchange.new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx = _hip.Context(0)
    try:
        ex = RcclExchange(ctx, world, rank, uid[0])
        got = ex.allgather_i64([rank * 10 + 1])
        print(f"rank {rank}: communicator ok, allgather -> {got.ravel().tolist()}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"rank {rank}: {type(exc).__name__}: {exc}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    mp.start_processes(rank_main, args=(2, 29671), nprocs=2, join=True, start_method="spawn")
