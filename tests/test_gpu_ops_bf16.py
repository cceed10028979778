"""-m gpu: the bf16 variants of the MFMA kernels (impl 3), kernel by kernel.  Inputs are rounded to
bfloat16 beforehand, so the in-register rounding is the identity, every product of two operands is exact
in float32 and only the summation order differs from torch-CPU float32 on the same rounded data: relative
max error <= 2e-5, like the float32 kernels.  With a load transform the kernel rounds AFTER BN+ReLU; the
oracle does the same."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import P, check, ctx, lib, nhwc, rel_err
from test_gpu_ops import CONV_SHAPES, CONVT_SHAPES

pytestmark = pytest.mark.gpu
BF = 3          # IMPL_MFMA_BF16: operands rounded to bf16 in registers (the transposed-conv kernels of the bf16 mode)
PBF = 6         # IMPL_PLANES_BF16: the plane kernels (bf16 activations staged by LDS-DMA), the 3x3 kernels of the bf16 mode
WSBF = 8        # IMPL_WS_BF16: the wave-specialised kernel with one plane (float32 tensors, operands rounded at staging)
TOL = 2e-5
SHAPES = [s for s in CONV_SHAPES if s[3] % 4 == 0 and s[4] % 4 == 0]
TSHAPES = [s for s in CONVT_SHAPES if s[3] % 4 == 0 and s[4] % 4 == 0]


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("xform", [False, True])
def test_conv3x3_forward_bf16(shape, xform):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = bf(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    if not xform:
        x = bf(x)
    xin = bf(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])) if xform else x
    want = nhwc(F.conv2d(xin.double(), wt.double(), b.double(), padding=1).float())
    c = ctx()
    dx, dw, db = c.to_device(nhwc(x)), c.to_device(wt.numpy()), c.to_device(b.numpy())
    dsc, dsh = c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in (BF, PBF) + ((WSBF,) if cin % 16 == 0 and h >= 8 and w >= 8 else ()):
        dy = c.empty((n, h, w, cout))
        check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout,
                                 P(dsc) if xform else None, P(dsh) if xform else None, 1 if xform else 0, P(dy)))
        assert rel_err(dy.numpy(), want) <= TOL, impl


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_dgrad_wgrad_bf16(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(7 + hash(shape) % 1000)
    x = bf(torch.randn(n, cin, h, w, generator=g)).double().requires_grad_(True)
    wt = bf(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).double().requires_grad_(True)
    dy = bf(torch.randn(n, cout, h, w, generator=g))
    F.conv2d(x, wt, None, padding=1).backward(dy.double())
    c = ctx()
    dxd, dwd, ddy = c.to_device(nhwc(x.detach().float())), c.to_device(wt.detach().float().numpy()), c.to_device(nhwc(dy))
    for impl in (BF, PBF):
        out = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dwd), cin, P(out)))
        assert rel_err(out.numpy(), nhwc(x.grad.float())) <= TOL, impl
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(dxd), P(ddy), n, h, w, cin, cout, None, None, 0, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.float().numpy()) <= 5e-5, impl
    if cout % 16 == 0 and h >= 8 and w >= 8:          # the input gradient on the wave-specialised kernel (Cin = cout)
        out = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, WSBF, P(ddy), n, h, w, cout, P(dwd), cin, P(out)))
        assert rel_err(out.numpy(), nhwc(x.grad.float())) <= TOL, "ws bf16"


@pytest.mark.parametrize("shape", TSHAPES)
def test_convt2x2_all_bf16(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(11 + hash(shape) % 1000)
    x = bf(torch.randn(n, cin, h, w, generator=g)).double().requires_grad_(True)
    wt = bf(torch.randn(cin, cout, 2, 2, generator=g) / (2 * cin ** 0.5)).double().requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose2d(x, wt, b.double(), stride=2)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.double())
    c = ctx()
    dx, dw, db, ddy = (c.to_device(nhwc(x.detach().float())), c.to_device(wt.detach().float().numpy()),
                       c.to_device(b.numpy()), c.to_device(nhwc(dy)))
    out = c.empty((n, 2 * h, 2 * w, cout))
    check(lib.rfi_op_convt2x2(c.handle, BF, P(dx), n, h, w, cin, P(dw), P(db), cout, P(out)))
    assert rel_err(out.numpy(), nhwc(y.detach().float())) <= TOL
    gx = c.empty((n, h, w, cin))
    check(lib.rfi_op_convt2x2_dgrad(c.handle, BF, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
    assert rel_err(gx.numpy(), nhwc(x.grad.float())) <= TOL
    gw = c.empty((cin, cout, 2, 2))
    check(lib.rfi_op_convt2x2_wgrad(c.handle, BF, P(dx), P(ddy), n, h, w, cin, cout, P(gw)))
    assert rel_err(gw.numpy(), wt.grad.float().numpy()) <= 5e-5


def test_wgrad_bf16_with_load_transform():
    n, h, w, cin, cout = 2, 16, 16, 32, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    a = bf(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])).double()
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    dy = bf(torch.randn(n, cout, h, w, generator=g))
    F.conv2d(a, wt, None, padding=1).backward(dy.double())
    c = ctx()
    dx, ddy, dsc, dsh = c.to_device(nhwc(x)), c.to_device(nhwc(dy)), c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in (BF, PBF):
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(dx), P(ddy), n, h, w, cin, cout, P(dsc), P(dsh), 1, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.float().numpy()) <= 5e-5, impl


# shapes that select the double-tile instantiations of the conv kernel (>= 512 workgroups; see test_gpu_ops.py)
from test_gpu_ops import BIG_SHAPES  # noqa: E402


@pytest.mark.parametrize("shape", BIG_SHAPES)
def test_conv3x3_double_tile_kernels_bf16(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(201 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = bf(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    dy = bf(torch.randn(n, cout, h, w, generator=g))
    xin = bf(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])).double().requires_grad_(True)
    y = F.conv2d(xin, wt.double(), b.double(), padding=1)
    y.backward(dy.double())
    c = ctx()
    dx, dw, db, ddy = c.to_device(nhwc(x)), c.to_device(wt.numpy()), c.to_device(b.numpy()), c.to_device(nhwc(dy))
    dsc, dsh = c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in (BF, PBF):
        out = c.empty((n, h, w, cout))
        check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout, P(dsc), P(dsh), 1, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach().float())) <= TOL, impl
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(xin.grad.float())) <= TOL, impl
