"""-m gpu: the RCCL gradient exchange on the one GPU this box has: a world of 1 goes through the
whole native path (dlopen librccl, ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the library's
stream), and a data-parallel step with world 1 must equal the plain step bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from rfi_toolbox_amd import distributed as D
from rfi_toolbox_amd._lib import Hyper, check, lib
from rfi_toolbox_amd.models import UNet
from rfi_toolbox_amd.runtime import Context

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one_allreduce_and_step():
    ctx = Context.get(0)
    uid = D.exchange_unique_id(ctx.comm_unique_id, 0, 1)
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(uid, 0, 1)
    try:
        x = np.arange(1000, dtype=np.float32) - 300.0
        d = ctx.to_device(x)
        check(lib.rfi_comm_allreduce_sum_f32(ctx.handle, C.c_void_p(d.ptr), x.size))
        ctx.synchronize()
        np.testing.assert_array_equal(d.numpy(), x)          # sum over a world of one

        g = torch.Generator().manual_seed(2)
        xb = torch.randn(2, 32, 32, 3, generator=g)
        yb = (torch.rand(2, 32, 32, generator=g) > 0.7).to(torch.uint8)
        torch.manual_seed(11)
        a = UNet(3, 1, 8)
        torch.manual_seed(11)
        b = UNet(3, 1, 8)
        la = a.train_step(xb, yb, lr=1e-3)
        lb = b.forward_backward(xb, yb)
        b.allreduce_gradients()                              # ncclAllReduce of the flat grad buffer
        b.apply_gradients(lr=1e-3, grad_scale=1.0)
        assert la == lb
        sa, sb = a.state_dict(), b.state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        # the async bench path picks the communicator up by itself
        hp = Hyper(1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0)
        dx, dy = ctx.to_device(xb.numpy()), ctx.to_device(yb.numpy())
        a.train_step_async(dx.ptr, dy.ptr, 2, 32, 32, hp)
        loss, norm = a.last_loss()
        assert np.isfinite(loss) and np.isfinite(norm)
    finally:
        ctx.comm_destroy()
