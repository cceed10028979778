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


def _flat_grads(model):
    p, n = C.c_void_p(), C.c_int64()
    check(lib.rfi_model_grad_buffer(model._h, C.byref(p), C.byref(n)))
    return p.value, n.value


def test_data_parallel_semantics_two_replicas_on_one_gpu():
    """the N-rank step is 'per-rank forward/backward with LOCAL BatchNorm statistics and
    local dice, gradients summed and scaled by 1/world, identical clip+Adam on every rank'.  With one
    GPU the two ranks are two replicas run in turn; the exchange is done by hand on the same flat
    gradient buffers RCCL all-reduces in place, and the result is checked against the CPU emulation
    'two micro-batches, grads averaged' built from the oracle."""
    from collections import OrderedDict

    from oracle import unet_ref
    ctx = Context.get(0)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 32, 32, 3, generator=g)
    y = (torch.rand(4, 32, 32, generator=g) > 0.75).to(torch.uint8)
    shards = [D.shard_range(4, r, 2) for r in range(2)]
    torch.manual_seed(5)
    reps = [UNet(3, 1, 8).train()]
    reps.append(UNet(3, 1, 8).load_state_dict(reps[0].state_dict()).train())
    st0 = reps[0].state_dict()
    losses = [m.forward_backward(x[lo:hi], y[lo:hi]) for m, (lo, hi) in zip(reps, shards)]
    ptrs = [_flat_grads(m) for m in reps]
    n = ptrs[0][1]
    host = []
    for p, _ in ptrs:
        h = np.empty(n, np.float32)
        check(lib.rfi_memcpy(ctx.handle, h.ctypes.data_as(C.c_void_p), 0, C.c_void_p(p), 1, h.nbytes))
        host.append(h)
    total = host[0] + host[1]                                  # == ncclAllReduce(sum)
    for p, _ in ptrs:
        check(lib.rfi_memcpy(ctx.handle, C.c_void_p(p), 1, total.ctypes.data_as(C.c_void_p), 0, total.nbytes))
    norms = [m.apply_gradients(lr=1e-3, grad_scale=0.5) for m in reps]
    assert norms[0] == norms[1]
    sa, sb = reps[0].state_dict(), reps[1].state_dict()
    for k in sa:                                               # replicas stay identical in the parameters
        if "running" not in k and "num_batches" not in k:
            assert torch.equal(sa[k], sb[k]), k

    # CPU emulation from the oracle
    gsum, bufs0, l_ref = None, None, []
    for r, (lo, hi) in enumerate(shards):
        l, _, gr, bufs = unet_ref.loss_and_grads(st0, unet_ref.nhwc_to_nchw(x[lo:hi]), y[lo:hi].float().unsqueeze(1))
        l_ref.append(float(l))
        gsum = gr if gsum is None else OrderedDict((k, gsum[k] + gr[k]) for k in gr)
        if r == 0:
            bufs0 = bufs
    avg = OrderedDict((k, v * 0.5) for k, v in gsum.items())
    total_norm, coef = unet_ref.clip_coefficient(avg, 1.0)
    assert losses == pytest.approx(l_ref, abs=2e-6)
    assert norms[0] == pytest.approx(float(total_norm), rel=2e-4)
    for k in ("encoder1.conv.conv.0.weight", "bottleneck.conv.3.weight", "decoder1.up.weight", "final_conv.weight"):
        want = avg[k].numpy()
        got = reps[0].grad(k) * 0.5                            # grad buffer holds the SUM; scale applied in Adam
        assert np.linalg.norm(got - want) <= 2e-3 * np.linalg.norm(want), k
    for k, v in bufs0.items():                                 # rank 0's BatchNorm buffers: local statistics
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(sa[k].numpy(), v.numpy(), rtol=0, atol=2e-6, err_msg=k)


# ------------------------------------------------------------------ the N-rank job as separate processes
def _launch_ddp(world, tmp_path, steps=2):
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), str(steps)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]


def _emulate(world, steps, lr=1e-3):
    """The oracle's statement of the N-rank step: per-rank loss/gradients with LOCAL BatchNorm statistics,
    gradients averaged, the reference's clip + Adam on the average; rank 0's BatchNorm buffers."""
    import importlib.util
    import os
    from collections import OrderedDict

    from oracle import unet_ref
    spec = importlib.util.spec_from_file_location("ddp_worker", os.path.join(os.path.dirname(__file__), "ddp_worker.py"))
    w = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w)
    x, y = w.batch()
    torch.manual_seed(5)
    st = UNet(3, 1, 8).state_dict()
    adam = unet_ref.new_adam_state(st)
    losses0, norms = [], []
    for _ in range(steps):
        gsum, bufs0 = None, None
        for r in range(world):
            lo, hi = D.shard_range(4, r, world)
            l, _, gr, bufs = unet_ref.loss_and_grads(st, unet_ref.nhwc_to_nchw(x[lo:hi]), y[lo:hi].float().unsqueeze(1))
            gsum = gr if gsum is None else OrderedDict((k, gsum[k] + gr[k]) for k in gr)
            if r == 0:
                bufs0, l0 = bufs, float(l)
        avg = OrderedDict((k, v / world) for k, v in gsum.items())
        total, coef = unet_ref.clip_coefficient(avg, 1.0)
        adam["step"] += 1
        t = adam["step"]
        bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
        with torch.no_grad():
            for k, gk in avg.items():
                gk = (gk * coef).add(st[k], alpha=1e-5)
                mm = adam["m"][k].lerp_(gk, 0.1)
                vv = adam["v"][k].mul_(0.999).addcmul_(gk, gk, value=0.001)
                st[k] = st[k].addcdiv(mm, (vv.sqrt() / bc2 ** 0.5).add_(1e-8), value=-(lr / bc1))
            for k, v in bufs0.items():
                st[k] = v
        losses0.append(l0)
        norms.append(float(total))
    return st, losses0, norms


def _check_against_emulation(ranks, world, steps):
    st, losses0, norms = _emulate(world, steps)
    r0 = ranks[0]
    assert r0["losses"][0] == pytest.approx(losses0[0], abs=2e-6)
    assert r0["norms"][0] == pytest.approx(norms[0], rel=2e-4)
    for k, v in st.items():
        if k.endswith("num_batches_tracked"):
            assert int(r0[k]) == int(v), k
        elif not (k.endswith(".0.bias") or k.endswith(".3.bias")):
            # two Adam steps at lr 1e-3: an element whose gradient is at rounding level may move either way
            d = np.abs(r0[k] - v.numpy())
            assert (d > 2e-4).mean() <= 2e-3 and d.max() <= 4.5e-3, (k, d.max(), (d > 2e-4).mean())
    for r in ranks[1:]:                                        # replicas stay bit-identical in the parameters
        for k in st:
            if "running" not in k and "num_batches" not in k:
                np.testing.assert_array_equal(r[k], r0[k], err_msg=k)
        np.testing.assert_array_equal(r["norms"], r0["norms"])


def test_ddp_worker_world_of_one(tmp_path):
    """The helper the N-rank test launches, run as a 1-rank job on this box's one GPU: its output must match the
    emulation with world = 1, i.e. the plain reference step (validates worker and checker without a second GPU)."""
    _check_against_emulation(_launch_ddp(1, tmp_path), 1, 2)


def test_rccl_two_ranks_public_train_step(tmp_path):
    """Two processes, two GPUs, RCCL over xGMI: the public `train_step` must all-reduce (it did not)
    and reproduce the oracle emulation 'two micro-batches, local BN, averaged gradients'."""
    n = C.c_int()
    check(lib.rfi_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("needs >= 2 GPUs in one box (the build/test pool has 1-GPU boxes)")
    _check_against_emulation(_launch_ddp(2, tmp_path), 2, 2)


@pytest.mark.parametrize("mode", ["float32", "bfloat16"])
def test_bucketed_exchange_emulation_is_bitwise_neutral(mode):
    """The bucketed, overlapped gradient exchange on ONE GPU (include/rfi_hip.h, rfi_comm_emulate): every bucket's
    all-reduce is replaced by 'multiply the range by 2' on the communication stream and the step applies
    grad_scale = 1/2.  Scaling by powers of two is exact, so weights, Adam moments and the reported gradient norm
    must equal the plain step bit for bit -- iff every element of the flat gradient buffer is exchanged exactly
    once, after its last producer (main and side streams) and before clip + Adam.  Three steps, so that a stale
    event or a bucket racing the next step's backward pass would show."""
    ctx = Context.get(0)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(4, 64, 64, 3, generator=g)
    y = (torch.rand(4, 64, 64, generator=g) > 0.7).to(torch.uint8)
    dx, dy = ctx.to_device(x.numpy()), ctx.to_device(y.numpy())
    hp = Hyper(1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0)
    out = []
    try:
        for world in (0, 2):
            ctx.comm_emulate(world)
            torch.manual_seed(23)
            m = UNet(3, 1, 16).set_compute_dtype(mode)
            stats = []
            for i in range(3):
                if i == 1:
                    m.train_step_async(dx.ptr, dy.ptr, 4, 64, 64, hp)        # both full-step entry points
                    stats.append(m.last_loss())
                else:
                    stats.append((m.train_step(x, y, lr=1e-3), m.last_loss()[1]))
            out.append((stats, m.state_dict(), m.adam_state("bottleneck.conv.3.weight"), m.adam_state("final_conv.bias")))
    finally:
        ctx.comm_emulate(0)
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
    for i in (2, 3):
        np.testing.assert_array_equal(out[0][i][0], out[1][i][0])
        np.testing.assert_array_equal(out[0][i][1], out[1][i][1])


@pytest.mark.parametrize("min_floats", [None, "1"])
@pytest.mark.parametrize("arch", ["resnet", "cnn3"])
def test_bucketed_exchange_emulation_other_architectures(arch, min_floats, monkeypatch):
    """The same bit-for-bit property for the models whose backward passes hand their gradient ranges to bucket_ready in
    another order (the ResNet-style encoder) or as one range (the 3-layer CNN), with the default merge threshold (buckets
    below 4 MB wait for their neighbours) and with RFI_BUCKET_MIN_FLOATS=1 (every bucket leaves alone): a range that is
    not adjacent to the pending one, or one that is never flushed, fails here and not at step 1 of an N-GPU job."""
    from rfi_toolbox_amd.models import SimpleCNN, UNetResNet18
    if min_floats is None:
        monkeypatch.delenv("RFI_BUCKET_MIN_FLOATS", raising=False)
    else:
        monkeypatch.setenv("RFI_BUCKET_MIN_FLOATS", min_floats)
    ctx = Context.get(0)
    g = torch.Generator().manual_seed(43)
    x = torch.randn(4, 64, 64, 3, generator=g)
    y = (torch.rand(4, 64, 64, generator=g) > 0.7).to(torch.uint8)
    out = []
    try:
        for world in (0, 2):
            ctx.comm_emulate(world)
            torch.manual_seed(29)
            m = UNetResNet18(3, 1, 16) if arch == "resnet" else SimpleCNN(3, 1, 32)
            stats = [(m.train_step(x, y, lr=1e-3), m.last_loss()[1]) for _ in range(3)]
            out.append((stats, m.state_dict()))
    finally:
        ctx.comm_emulate(0)
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
