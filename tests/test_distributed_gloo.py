"""CPU-only, world_size 2 over gloo: the control plane and the data-parallel semantics of
rfi_toolbox_amd.distributed (rendezvous on 127.0.0.1, unique-id broadcast, balanced shards,
max-over-ranks timing) and the definition the RCCL exchange implements -- mean of per-rank
gradients with rank-local BatchNorm/dice -- checked with the CPU oracle as the per-rank worker.
No RCCL and no GPU here; the exchange itself (rfi_comm_allreduce_sum_f32) runs on the GPU box."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from rfi_toolbox_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from oracle import unet_ref
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, lr, w = D.init_control_plane("gloo")
    assert (r, lr, w) == (rank, rank, world)
    uid = D.exchange_unique_id(lambda: bytes(range(128)), rank, world)      # stand-in for ncclGetUniqueId
    assert uid == bytes(range(128))
    # identical replicas, disjoint shards of a global batch of 6 patches
    st = unet_ref.init_state(3, 1, 4, seed=7)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 3, 16, 16, generator=g)
    y = (torch.rand(6, 1, 16, 16, generator=g) > 0.7).float()
    lo, hi = D.shard_range(6, rank, world)
    _, _, grads, _ = unet_ref.loss_and_grads(st, x[lo:hi], y[lo:hi])
    flat = torch.cat([v.reshape(-1) for v in grads.values()])
    dist.all_reduce(flat)                       # what ncclAllReduce(sum) does to the flat buffer
    flat /= world                               # grad_scale = 1/world in rfi_train_apply
    assert D.max_over_ranks(float(rank)) == world - 1
    D.barrier()
    np.save(os.path.join(out_dir, f"avg_{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


def test_world2_gradient_averaging(tmp_path):
    from oracle import unet_ref
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a0, a1 = np.load(tmp_path / "avg_0.npy"), np.load(tmp_path / "avg_1.npy")
    assert np.array_equal(a0, a1)                                    # every rank applies the same update
    # single-process emulation: 2 micro-batches, local BN statistics / dice, gradients averaged
    st = unet_ref.init_state(3, 1, 4, seed=7)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 3, 16, 16, generator=g)
    y = (torch.rand(6, 1, 16, 16, generator=g) > 0.7).float()
    per_rank = []
    for r in range(2):
        lo, hi = D.shard_range(6, r, 2)
        _, _, grads, _ = unet_ref.loss_and_grads(st, x[lo:hi], y[lo:hi])
        per_rank.append(torch.cat([v.reshape(-1) for v in grads.values()]).numpy())
    np.testing.assert_allclose(a0, D.average_gradients_reference(per_rank), rtol=0, atol=2e-6)
    # and it is NOT the full-batch gradient (BatchNorm/dice are batch-global in one process)
    _, _, gfull, _ = unet_ref.loss_and_grads(st, x, y)
    full = torch.cat([v.reshape(-1) for v in gfull.values()]).numpy()
    assert np.abs(full - a0).max() > 1e-5


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 5, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def test_single_process_paths_need_no_process_group():
    assert D.exchange_unique_id(lambda: b"x" * 128, 0, 1) == b"x" * 128
    assert D.max_over_ranks(3.5) == 3.5
    D.barrier()
