"""Helper of the data-parallel tests (NOT collected by pytest): one rank of an N-rank job.

    RANK=r LOCAL_RANK=r WORLD_SIZE=N MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/ddp_worker.py OUT_DIR [STEPS]

Every rank builds the same UNet(3,1,8) (same torch seed), takes its shard of the same seeded 4-patch batch, and
runs the PUBLIC training step (`model.train_step` -> rfi_train_step), which all-reduces the gradients over RCCL
when the context holds a communicator.  Each rank writes OUT_DIR/rank<r>.npz (losses, grad norms, final state).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def batch():
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 32, 32, 3, generator=g)
    y = (torch.rand(4, 32, 32, generator=g) > 0.75).to(torch.uint8)
    return x, y


def main():
    out_dir = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from rfi_toolbox_amd import distributed as D
    from rfi_toolbox_amd.models import UNet
    from rfi_toolbox_amd.runtime import Context
    rank, local_rank, world = D.init_control_plane("gloo")
    ctx = Context.get(local_rank)
    D.init_gradient_exchange(ctx, rank, world)
    torch.manual_seed(5)
    m = UNet(3, 1, 8, device=local_rank).train()
    x, y = batch()
    lo, hi = D.shard_range(4, rank, world)
    losses, norms = [], []
    for _ in range(steps):
        losses.append(m.train_step(x[lo:hi], y[lo:hi], lr=1e-3))
        norms.append(m.last_loss()[1])
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array(losses), norms=np.array(norms), **sd)
    D.barrier()
    ctx.comm_destroy()


if __name__ == "__main__":
    main()
