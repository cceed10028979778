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


# ------------------------------------------------------------------ elementwise kernels of the bfloat16 data flow
def _bits(t):          # float32 tensor holding bf16 values -> uint16 bit patterns (numpy)
    import numpy as np
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def _from_bits(a):
    import numpy as np
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("m,c", [(4096, 64), (1000, 16), (333, 128), (64 * 64 * 3, 32)])
@pytest.mark.parametrize("projection", [False, True])
def test_block_tail_bf16(m, c, projection):
    """relu(BN(y) + shortcut) on bfloat16 tensors: float32 arithmetic (product and sum rounded separately), ONE rounding of
    the result to bfloat16 -- bit-exact against the same expression in torch."""
    import numpy as np
    g = torch.Generator().manual_seed(m + c)
    y, s = bf(torch.randn(m, c, generator=g)), bf(torch.randn(m, c, generator=g))
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    ssc, ssh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    short = (s * ssc + ssh) if projection else s
    want = torch.relu((y * sc + sh) + short).to(torch.bfloat16).to(torch.float32)
    cx = ctx()
    dy, ds = cx.to_device(_bits(y)), cx.to_device(_bits(s))
    dsc, dsh, dssc, dssh = (cx.to_device(t.numpy()) for t in (sc, sh, ssc, ssh))
    out = cx.empty((m, c), np.uint16)
    check(lib.rfi_op_bn_add_relu16(cx.handle, P(dy), P(dsc), P(dsh), P(ds), P(dssc) if projection else None, P(dssh) if projection else None,
                                   m, c, P(out)))
    assert torch.equal(_from_bits(out.numpy()), want)


@pytest.mark.parametrize("m,c", [(4096, 64), (777, 16), (64 * 64, 256)])
@pytest.mark.parametrize("terms", ["g0", "g0+g1", "g0+f32", "g0+g1+bf16", "nomask"])
def test_masked_gradient_sum_bf16(m, c, terms):
    """dz = (g0 + g1 + g2) * [a > 0]: the gradient terms that meet at a BasicBlock output (the next block's input gradient, the
    identity shortcut's, the decoder's skip gradient as a channel slice of a float32 or bfloat16 [up | skip] tensor), summed in
    float32 in that order and rounded once -- bit-exact against torch."""
    import numpy as np
    g = torch.Generator().manual_seed(m * 3 + c)
    g0, g1, a = (bf(torch.randn(m, c, generator=g)) for _ in range(3))
    cat = torch.randn(m, 2 * c, generator=g)
    cat16 = bf(cat)
    tot = g0.clone()
    if "g1" in terms:
        tot = tot + g1
    if "f32" in terms:
        tot = tot + cat[:, c:]
    if "bf16" in terms:
        tot = tot + cat16[:, c:]
    want = (tot if terms == "nomask" else torch.where(a > 0, tot, torch.zeros_like(tot))).to(torch.bfloat16).to(torch.float32)
    cx = ctx()
    d0, d1, da = cx.to_device(_bits(g0)), cx.to_device(_bits(g1)), cx.to_device(_bits(a))
    dcat, dcat16 = cx.to_device(cat.numpy()), cx.to_device(_bits(cat16))
    out = cx.empty((m, c), np.uint16)
    import ctypes as C
    f32p = C.c_void_p(dcat.ptr + 4 * c) if "f32" in terms else None
    b16p = C.c_void_p(dcat16.ptr + 2 * c) if "bf16" in terms else None
    check(lib.rfi_op_relu_mask_sum16(cx.handle, P(d0), P(d1) if "g1" in terms else None, f32p, b16p, 2 * c,
                                     None if terms == "nomask" else P(da), m, c, P(out)))
    assert torch.equal(_from_bits(out.numpy()), want)


@pytest.mark.parametrize("m,c", [(64 * 64 * 4, 32), (128 * 128, 64), (3000, 16), (8 * 8 * 64, 512)])
@pytest.mark.parametrize("slope", [0.0, 1.0, 0.1])
def test_batchnorm_backward_bf16(m, c, slope):
    """Training-mode BatchNorm backward in front of a (Leaky)ReLU on bfloat16 tensors (8 channels per lane, float32 partial sums
    per 8 rows, fp64 across): dgamma / dbeta / the conv-bias gradient and dY against float64 torch on the same bf16 inputs."""
    import numpy as np
    g = torch.Generator().manual_seed(m + 7 * c)
    y, da = bf(torch.randn(m, c, generator=g) * 1.5 + 0.2), bf(torch.randn(m, c, generator=g))
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    y64 = y.double()
    mean, var = y64.mean(0), y64.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - mean * gamma.double() * invstd).float()
    mean32, invstd32 = mean.float(), invstd.float()
    z = y * scale + shift                                   # the forward's own float32 pre-activation decides the mask
    dz = torch.where(z > 0, da, da * slope).double()
    xh = ((y - mean32) * invstd32).double()
    dbeta, dgamma = dz.sum(0), (dz * xh).sum(0)
    dyw = (gamma * invstd32).double() * (dz - dbeta / m - xh * (dgamma / m))
    cx = ctx()
    dda, dyy = cx.to_device(_bits(da)), cx.to_device(_bits(y))
    dev = {k: cx.to_device(v.numpy()) for k, v in dict(gamma=gamma, scale=scale, shift=shift, mean=mean32, invstd=invstd32).items()}
    out = cx.empty((m, c), np.uint16)
    og, ob, obias = cx.empty((c,)), cx.empty((c,)), cx.empty((c,))
    check(lib.rfi_op_bn_backward16(cx.handle, P(dda), P(dyy), m, c, P(dev["gamma"]), P(dev["scale"]), P(dev["shift"]), P(dev["mean"]),
                                   P(dev["invstd"]), slope, P(out), P(og), P(ob), P(obias)))
    got = _from_bits(out.numpy()).double()
    assert float((got - dyw).abs().max()) <= 2.0 ** -8 * float(dyw.abs().max()) + 1e-6
    assert rel_err(og.numpy(), dgamma.float().numpy()) <= 2e-5 and rel_err(ob.numpy(), dbeta.float().numpy()) <= 2e-5
    # the conv bias in front of the BatchNorm: sum of the (rounded) dY, ~0 by construction
    assert float(np.abs(obias.numpy() - got.sum(0).float().numpy()).max()) <= 1e-3 * float(got.abs().sum(0).max()) + 1e-6
