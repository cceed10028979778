"""Data-parallel training across the GPUs of one node: one process per GPU, identical weight
replicas, per-rank shards of the patches, ONE exchange per step -- an RCCL all-reduce(sum) of
the flat fp32 gradient buffer over xGMI issued by librfi_hip.so on the context's own HIP stream,
followed by the clip+Adam kernel with grad_scale = 1/world (gradient averaging).  The reference
has no multi-GPU path; semantics are torch-DDP-like: BatchNorm statistics and the
dice term are local to each rank's micro-batch, gradients are averaged.

torch.distributed (gloo) is only the control plane: rendezvous, broadcast of the 128-byte
ncclUniqueId, barriers and the max-over-ranks of timings.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) shard of n_items for `rank` (first n%world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_control_plane(backend="gloo"):
    """Join the torch.distributed rendezvous described by RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT.  The rendezvous and
    every later control-plane collective time out after RFI_RDZV_TIMEOUT seconds (default 180) instead of waiting
    forever for a rank that never came up."""
    import datetime

    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        timeout = datetime.timedelta(seconds=float(os.environ.get("RFI_RDZV_TIMEOUT", "180")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def exchange_unique_id(make_id, rank: int, world: int) -> bytes:
    """Rank 0 calls make_id() -> 128 bytes; every rank returns the same bytes."""
    if world == 1:
        return make_id()
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("ncclUniqueId exchange failed")
    return bytes(uid)


def init_gradient_exchange(ctx, rank: int, world: int):
    """Create the RCCL communicator of `ctx` (no-op for world == 1)."""
    if world == 1:
        return
    uid = exchange_unique_id(ctx.comm_unique_id, rank, world)
    ctx.comm_init(uid, rank, world)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def shutdown():
    """Leave the control plane in an orderly way: a last barrier, then destroy the process group (a rank that simply exits
    while its peers still hold connections to it can take them down in their teardown)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def count_ranks() -> int:
    """Ranks the control plane really connects: a sum of ones (1 without a process group)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t[0])
    return 1


def count_ranks_rccl(ctx, world: int) -> int:
    """Ranks the RCCL communicator of `ctx` connects: an all-reduce(sum) of one float over it (1 for world == 1)."""
    if world == 1:
        return 1
    import ctypes as C

    from ._lib import check, lib
    one = ctx.to_device(np.ones(4, np.float32))
    check(lib.rfi_comm_allreduce_sum_f32(ctx.handle, C.c_void_p(one.ptr), 4))
    ctx.synchronize()
    return int(round(float(one.numpy()[0])))


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])
    return value


def average_gradients_reference(grads_per_rank):
    """What the exchange computes, stated on host arrays (used by the CPU tests): mean over ranks."""
    return np.mean(np.stack([np.asarray(g, dtype=np.float32) for g in grads_per_rank]), axis=0,
                   dtype=np.float32)
