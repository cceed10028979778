"""-m gpu: every HIP conv-like kernel (direct VALU and MFMA implicit GEMM) against torch-CPU fp32
(F.conv2d / F.conv_transpose2d + autograd, the substrate the reference runs on) on seeded inputs.
Tolerance: fp32 path, relative max error <= 2e-5 of the output scale (fp32 MFMA is an exact fmaf
chain; the difference is summation order only)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import IMPL_DIRECT, IMPL_MFMA, P, check, ctx, lib, nchw, nhwc, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5

# (n, h, w, cin, cout): stem-like, tiny channels, every MFMA tile family, ragged sizes
CONV_SHAPES = [
    (2, 16, 16, 3, 8), (1, 8, 8, 4, 4), (2, 32, 32, 32, 32), (2, 64, 64, 32, 64), (1, 16, 16, 64, 128),
    (2, 8, 8, 128, 64), (1, 24, 40, 16, 48), (3, 4, 4, 8, 16), (1, 128, 128, 32, 32), (1, 12, 20, 20, 36),
]


def _impls(cin):
    return [IMPL_DIRECT, IMPL_MFMA] if cin % 4 == 0 else [IMPL_DIRECT]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("xform", [False, True])
def test_conv3x3_forward(shape, xform):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    xin = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]) if xform else x
    want = nhwc(F.conv2d(xin, wt, b, padding=1))
    c = ctx()
    dx, dw, db = c.to_device(nhwc(x)), c.to_device(wt.numpy()), c.to_device(b.numpy())
    dsc, dsh = c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in _impls(cin):
        dy = c.empty((n, h, w, cout))
        check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout,
                                 P(dsc) if xform else None, P(dsh) if xform else None, 1 if xform else 0, P(dy)))
        assert rel_err(dy.numpy(), want) <= TOL, f"impl={impl}"


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_dgrad_wgrad(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(7 + hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).requires_grad_(True)
    dy = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    c = ctx()
    dxd, dwd, ddy = c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(nhwc(dy))
    for impl in _impls(cin) if cout % 4 == 0 else [IMPL_DIRECT]:
        out = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dwd), cin, P(out)))
        assert rel_err(out.numpy(), nhwc(x.grad)) <= TOL, f"dgrad impl={impl}"
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(dxd), P(ddy), n, h, w, cin, cout, None, None, 0, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"wgrad impl={impl}"


def test_conv3x3_wgrad_with_load_transform():
    n, h, w, cin, cout = 2, 16, 16, 32, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    a = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    dy = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(a, wt, None, padding=1).backward(dy)
    c = ctx()
    for impl in (IMPL_DIRECT, IMPL_MFMA):
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(c.to_device(nhwc(x))), P(c.to_device(nhwc(dy))), n, h, w,
                                       cin, cout, P(c.to_device(sc.numpy())), P(c.to_device(sh.numpy())), 1, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"impl={impl}"


CONVT_SHAPES = [(2, 8, 8, 64, 32), (1, 4, 4, 16, 8), (2, 16, 16, 128, 64), (1, 32, 32, 64, 32), (2, 2, 2, 8, 4),
                (1, 6, 10, 12, 20)]


@pytest.mark.parametrize("shape", CONVT_SHAPES)
def test_convt2x2_all(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(11 + hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cin, cout, 2, 2, generator=g) / (2 * cin ** 0.5)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose2d(x, wt, b, stride=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    c = ctx()
    dx, dw, db, ddy = (c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(b.numpy()),
                       c.to_device(nhwc(dy)))
    for impl in (IMPL_DIRECT, IMPL_MFMA):
        out = c.empty((n, 2 * h, 2 * w, cout))
        check(lib.rfi_op_convt2x2(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL, f"fwd impl={impl}"
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_convt2x2_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(x.grad)) <= TOL, f"dgrad impl={impl}"
        gw = c.empty((cin, cout, 2, 2))
        check(lib.rfi_op_convt2x2_wgrad(c.handle, impl, P(dx), P(ddy), n, h, w, cin, cout, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"wgrad impl={impl}"


@pytest.mark.parametrize("m,c_", [(1, 4), (37, 6), (4096, 32), (70000, 64), (513, 200)])
def test_bn_statistics(m, c_):
    g = torch.Generator().manual_seed(m)
    y = torch.randn(m, c_, generator=g) * (torch.rand(c_, generator=g) * 3 + 0.01) + torch.randn(c_, generator=g) * 50
    c = ctx()
    mean, var = c.empty((c_,)), c.empty((c_,))
    check(lib.rfi_op_bn_stats(c.handle, P(c.to_device(y.numpy())), m, c_, P(mean), P(var)))
    yd = y.double()
    np.testing.assert_allclose(mean.numpy(), yd.mean(0).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(var.numpy(), yd.var(0, unbiased=False).numpy(), rtol=2e-5, atol=1e-7)
