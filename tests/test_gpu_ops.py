"""-m gpu: every HIP conv-like kernel (direct VALU, MFMA implicit GEMM in native float32 and in the default
float32-by-3xbf16 arithmetic) against torch-CPU fp32
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
    # the stem shapes (Cin = 4 = three channels padded, Cout 32 / 64): conv_stem.hip / wgrad_stem.hip, whole and ragged tiles
    (2, 16, 32, 4, 32), (1, 13, 37, 4, 32), (3, 8, 16, 4, 64), (1, 40, 24, 4, 64),
]


IMPL_X3 = 4       # MFMA kernel, float32 by 3 x bf16 splitting (the default arithmetic of the models)
IMPL_PX3 = 5      # plane kernels (LDS-DMA staged bf16 pieces), 3 x bf16 arithmetic; any channel count
IMPL_WS = 7       # wave-specialised kernel (producer waves split, consumer waves multiply), 3 x bf16; H, W >= 8, Cin % 16 == 0


def _impls(cin, h=0, w=0, cout=0, xform=False):
    ws = [IMPL_WS] if (cin % 16 == 0 and h >= 8 and w >= 8) or (cin == 4 and cout in (32, 64) and not xform) else []
    return [IMPL_DIRECT, IMPL_MFMA, IMPL_X3, IMPL_PX3] + ws if cin % 4 == 0 else [IMPL_DIRECT, IMPL_PX3]


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
    for impl in _impls(cin, h, w, cout, xform):
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
    for impl in _impls(cin) if cout % 4 == 0 else [IMPL_DIRECT, IMPL_PX3]:
        out = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dwd), cin, P(out)))
        assert rel_err(out.numpy(), nhwc(x.grad)) <= TOL, f"dgrad impl={impl}"
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(dxd), P(ddy), n, h, w, cin, cout, None, None, 0, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"wgrad impl={impl}"
    if cout % 16 == 0 and h >= 8 and w >= 8:      # the input gradient on the wave-specialised kernel (a conv with Cin = cout)
        out = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, IMPL_WS, P(ddy), n, h, w, cout, P(dwd), cin, P(out)))
        assert rel_err(out.numpy(), nhwc(x.grad)) <= TOL, "dgrad impl=ws"


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
    dx, ddy, dsc, dsh = (c.to_device(nhwc(x)), c.to_device(nhwc(dy)), c.to_device(sc.numpy()),
                         c.to_device(sh.numpy()))      # keep the device buffers alive across the calls
    for impl in (IMPL_DIRECT, IMPL_MFMA, IMPL_X3, IMPL_PX3):
        gw = c.empty((cout, cin, 3, 3))
        check(lib.rfi_op_conv3x3_wgrad(c.handle, impl, P(dx), P(ddy), n, h, w, cin, cout, P(dsc), P(dsh), 1, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"impl={impl}"


CONVT_SHAPES = [(2, 8, 8, 64, 32), (1, 4, 4, 16, 8), (2, 16, 16, 128, 64), (1, 32, 32, 64, 32), (2, 2, 2, 8, 4),
                (1, 6, 10, 12, 20), (3, 5, 7, 48, 96), (64, 8, 8, 512, 256), (1, 3, 3, 32, 16),
                (2, 9, 12, 32, 64), (1, 7, 9, 64, 64), (2, 5, 6, 16, 32)]     # (every tile form of the stride-2 weight-gradient kernel, ragged)


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
    for impl in (IMPL_DIRECT, IMPL_MFMA, IMPL_X3) if cin % 4 == 0 and cout % 4 == 0 else (IMPL_DIRECT,):
        out = c.empty((n, 2 * h, 2 * w, cout))
        check(lib.rfi_op_convt2x2(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL, f"fwd impl={impl}"
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_convt2x2_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(x.grad)) <= TOL, f"dgrad impl={impl}"
        gw = c.empty((cin, cout, 2, 2))
        check(lib.rfi_op_convt2x2_wgrad(c.handle, impl, P(dx), P(ddy), n, h, w, cin, cout, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"wgrad impl={impl}"
    # the wave-specialised GEMM kernel: forward with its four phases folded into the channels, input gradient as four taps
    if cin % 16 == 0 and cout % 32 == 0:
        out = c.empty((n, 2 * h, 2 * w, cout))
        check(lib.rfi_op_convt2x2(c.handle, IMPL_WS, P(dx), n, h, w, cin, P(dw), P(db), cout, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL, "fwd impl=ws"
    if cout % 16 == 0:
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_convt2x2_dgrad(c.handle, IMPL_WS, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(x.grad)) <= TOL, "dgrad impl=ws"


@pytest.mark.parametrize("m,c_", [(1, 4), (37, 6), (4096, 32), (70000, 64), (513, 200)])
def test_bn_statistics(m, c_):
    g = torch.Generator().manual_seed(m)
    y = torch.randn(m, c_, generator=g) * (torch.rand(c_, generator=g) * 3 + 0.01) + torch.randn(c_, generator=g) * 50
    c = ctx()
    mean, var, dy = c.empty((c_,)), c.empty((c_,)), c.to_device(y.numpy())
    check(lib.rfi_op_bn_stats(c.handle, P(dy), m, c_, P(mean), P(var)))
    yd = y.double()
    np.testing.assert_allclose(mean.numpy(), yd.mean(0).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(var.numpy(), yd.var(0, unbiased=False).numpy(), rtol=2e-5, atol=1e-7)


def _unet_conv_shapes(f, n, size, depth=4):
    """(n, h, w, cin, cout) of every 3x3 conv of UNet(3,1,f) on n x size x size inputs."""
    out, cin = [], 3
    for lvl in range(depth):
        c = f << lvl
        s = size >> lvl
        out += [(n, s, s, cin, c), (n, s, s, c, c)]
        cin = c
    s = size >> depth
    out += [(n, s, s, cin, 2 * cin), (n, s, s, 2 * cin, 2 * cin)]
    for lvl in range(depth - 1, -1, -1):
        c = f << lvl
        s = size >> lvl
        out += [(n, s, s, 2 * c, c), (n, s, s, c, c)]
    return sorted(set(out))


@pytest.mark.parametrize("shape", _unet_conv_shapes(16, 2, 64) + _unet_conv_shapes(8, 1, 32))
def test_every_unet_layer_shape(shape):
    """forward, dgrad and wgrad at exactly the shapes the model launches (auto implementation)."""
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 997)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv2d(x, wt, b, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    c = ctx()
    dx, dw, db, ddy = (c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(b.numpy()),
                       c.to_device(nhwc(dy)))
    out = c.empty((n, h, w, cout))
    check(lib.rfi_op_conv3x3(c.handle, 0, P(dx), n, h, w, cin, P(dw), P(db), cout, None, None, 0, P(out)))
    assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL
    gx = c.empty((n, h, w, cin))
    check(lib.rfi_op_conv3x3_dgrad(c.handle, 0, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
    assert rel_err(gx.numpy(), nhwc(x.grad)) <= TOL
    gw = c.empty((cout, cin, 3, 3))
    check(lib.rfi_op_conv3x3_wgrad(c.handle, 0, P(dx), P(ddy), n, h, w, cin, cout, None, None, 0, P(gw)))
    assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5


POOL_SHAPES = [(2, 32, 32, 32), (2, 64, 64, 16), (1, 8, 8, 128), (3, 4, 6, 5), (1, 16, 16, 64)]


@pytest.mark.parametrize("shape", POOL_SHAPES)
def test_bn_relu_pool_and_backward(shape):
    n, h, w, ch = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.randn(n, ch, h, w, generator=g)
    y[:, :, ::4, ::4] = -3.0                               # force all-zero windows / ties at zero
    sc = torch.rand(ch, generator=g) + 0.5
    sc[::3] *= -1                                          # negative scales: max must follow the activation
    sh = torch.randn(ch, generator=g) * 0.2
    a = torch.relu(y * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    p = F.max_pool2d(a, 2, 2)
    dpool = torch.randn(p.shape, generator=g)
    dskip = torch.randn(a.shape, generator=g)
    (p * dpool).sum().backward()
    want_da = a.grad + dskip
    c = ctx()
    dy_, dsc, dsh = c.to_device(nhwc(y)), c.to_device(sc.numpy()), c.to_device(sh.numpy())
    skip, pooled = c.empty((n, h, w, ch)), c.empty((n, h // 2, w // 2, ch))
    check(lib.rfi_op_bn_relu_pool(c.handle, P(dy_), n, h, w, ch, P(dsc), P(dsh), P(skip), P(pooled)))
    np.testing.assert_allclose(skip.numpy(), nhwc(a.detach()), rtol=0, atol=1e-6)
    np.testing.assert_allclose(pooled.numpy(), nhwc(p.detach()), rtol=0, atol=1e-6)
    da = c.empty((n, h, w, ch))
    dsk, dpl = c.to_device(nhwc(dskip)), c.to_device(nhwc(dpool))
    check(lib.rfi_op_pool_bwd_merge(c.handle, P(dy_), n, h, w, ch, P(dsc), P(dsh), P(dsk), P(dpl), P(da)))
    np.testing.assert_allclose(da.numpy(), nhwc(want_da), rtol=0, atol=1e-6)


@pytest.mark.parametrize("m,ch", [(2048, 32), (8192, 16), (32, 128), (100, 6), (65536, 64)])
def test_bn_relu_backward(m, ch):
    g = torch.Generator().manual_seed(m + ch)
    y = (torch.randn(m, ch, generator=g) * (torch.rand(ch, generator=g) + 0.2) + torch.randn(ch, generator=g)).double()
    gamma = (torch.rand(ch, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(ch, generator=g) * 0.3).double().requires_grad_(True)
    da = torch.randn(m, ch, generator=g).double()
    yl = y.clone().requires_grad_(True)
    mean, var = yl.mean(0), yl.var(0, unbiased=False)
    act = torch.relu((yl - mean) / torch.sqrt(var + 1e-5) * gamma + beta)
    (act * da).sum().backward()
    c = ctx()
    dda = c.to_device(da.float().numpy())
    dy_, dg, db_ = c.to_device(y.float().numpy()), c.to_device(gamma.detach().float().numpy()), c.to_device(beta.detach().float().numpy())
    og, ob, obias = c.empty((ch,)), c.empty((ch,)), c.empty((ch,))
    check(lib.rfi_op_bn_relu_backward(c.handle, P(dy_), m, ch, P(dg), P(db_), P(dda), P(og), P(ob), P(obias)))
    scale = float(yl.grad.abs().max())
    bad = np.abs(dda.numpy() - yl.grad.numpy()) > 2e-5 * scale + 1e-7
    assert bad.sum() <= 2, int(bad.sum())       # a ReLU input within fp32 rounding of 0 may flip its mask
    for got, want in ((og.numpy(), gamma.grad.numpy()), (ob.numpy(), beta.grad.numpy())):
        off = np.abs(got - want) > 1e-4 * np.abs(want) + 1e-4
        assert off.sum() <= 2 and np.abs(got - want).max() <= 2 * float(da.abs().max()) * 6, off.sum()
    assert np.abs(obias.numpy()).max() <= 1e-3 * max(1.0, scale * m ** 0.5)


# ------------------------------------------------------------------ double-tile instantiations
# conv_igemm_kernel<R,S,16,32,32,4,1>, <8,32,64,4,1>, <16,16,64,4,1> are only selected when a launch has >= 512
# workgroups of the double tile (dispatch_tiles in conv_mfma.hip) and the arithmetic is native float32 or bf16:
# these shapes reach them (forward; the dgrad of the mirrored shape reaches them with Cin/Cout swapped).
BIG_SHAPES = [(16, 128, 128, 32, 32), (8, 128, 128, 32, 64), (64, 16, 16, 128, 512), (8, 128, 128, 64, 32)]


@pytest.mark.parametrize("shape", BIG_SHAPES)
def test_conv3x3_double_tile_kernels(shape):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(101 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    dy = torch.randn(n, cout, h, w, generator=g)
    xin = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    y = F.conv2d(xin, wt, b, padding=1)
    y.backward(dy)
    c = ctx()
    dx, dw, db, ddy = c.to_device(nhwc(x)), c.to_device(wt.numpy()), c.to_device(b.numpy()), c.to_device(nhwc(dy))
    dsc, dsh = c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in (IMPL_MFMA, IMPL_X3, IMPL_PX3):
        out = c.empty((n, h, w, cout))
        check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout, P(dsc), P(dsh), 1, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL, f"fwd impl={impl}"
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv3x3_dgrad(c.handle, impl, P(ddy), n, h, w, cout, P(dw), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(xin.grad)) <= TOL, f"dgrad impl={impl}"


# ------------------------------------------------------------------ adversarial inputs of the 3 x bf16 split
def _wide(shape, g, lo, hi):
    """randn scaled by 2^k, k uniform in [lo, hi]: a dynamic range no randn draw has."""
    k = torch.randint(lo, hi + 1, shape, generator=g).double()
    return (torch.randn(shape, generator=g).double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), k)).float()


@pytest.mark.parametrize("impl", [IMPL_MFMA, IMPL_X3, IMPL_PX3])
def test_conv3x3_wide_dynamic_range_and_cancellation(impl):
    """The float32-by-3xbf16 arithmetic claims ONE float32 rounding per product (the three dropped piece
    products are <= 2^-24 |a b|), for any finite operands, not just randn.  Operands spanning 2^-60 .. 2^60
    and channel pairs that cancel EXACTLY in exact arithmetic (x equal, weights opposite): the error against
    float64 is bounded per output by a few float32 roundings of sum |a||b| -- the same bound the native
    float32 MFMA (an fmaf chain) is held to."""
    n, h, w, cin, cout = 2, 16, 16, 32, 32
    g = torch.Generator().manual_seed(77)
    x = _wide((n, cin, h, w), g, -60, 60)
    wt = _wide((cout, cin, 3, 3), g, -30, 30)
    x[:, 1::2] = x[:, 0::2]                       # channel pairs (2k, 2k+1): same input ...
    wt[:, 1::2] = -wt[:, 0::2]                    # ... opposite weights: every pair cancels exactly
    x[0, :, 5, 5] = _wide((cin,), g, -60, 60)     # except where the pairing is broken at one pixel
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)        # sum |a||b|
    cancel = torch.ones(n, cout, h, w, dtype=torch.bool)                          # exact result is 0 there
    cancel[0, :, 4:7, 4:7] = False
    want[cancel] = 0.0                            # (float64 summation leaves ~2^-53 residues of its own)
    c = ctx()
    dx, dw = c.to_device(nhwc(x)), c.to_device(wt.numpy())
    out = c.empty((n, h, w, cout))
    check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), None, cout, None, None, 0, P(out)))
    got = torch.from_numpy(out.numpy()).permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    err = ((got - want).abs() / (bound + 1e-300)).max().item()
    # K = 288 products: an fmaf chain may lose up to K * 2^-24 of sum|ab| in the worst case, a few 2^-24 typically
    assert err <= 16 * 2.0 ** -24, err
    # where everything cancels the result must be tiny against the operands, not a stale piece product
    assert (got[cancel].abs() <= 16 * 2.0 ** -24 * bound[cancel]).all()


@pytest.mark.parametrize("impl", [IMPL_MFMA, IMPL_X3, IMPL_PX3])
def test_conv3x3_non_finite_lanes(impl):
    """An inf and a NaN in the input: every output whose 3x3 window contains one of them must come out
    non-finite, every other output must be unaffected.  (The split path turns inf into NaN -- inf - inf in the
    residual -- so 'non-finite' is the contract, not the kind; documented in DESIGN.md.)"""
    n, h, w, cin, cout = 1, 16, 16, 16, 32
    g = torch.Generator().manual_seed(78)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 12
    clean = F.conv2d(x, wt, None, padding=1)
    x[0, 3, 4, 4] = float("inf")
    x[0, 9, 10, 12] = float("nan")
    c = ctx()
    dx, dw = c.to_device(nhwc(x)), c.to_device(wt.numpy())
    out = c.empty((n, h, w, cout))
    check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), None, cout, None, None, 0, P(out)))
    got = out.numpy()[0]                                      # (h, w, cout)
    bad = np.zeros((h, w), bool)
    bad[3:6, 3:6] = True
    bad[9:12, 11:14] = True
    assert not np.isfinite(got[bad]).any()
    ok = ~bad
    assert np.isfinite(got[ok]).all()
    assert rel_err(got[ok], nhwc(clean)[0][ok]) <= TOL


@pytest.mark.parametrize("impl", [IMPL_MFMA, IMPL_X3, IMPL_PX3])
def test_conv3x3_tiny_operands_underflow(impl):
    """Operands near the bottom of the float32 range: the low bf16 pieces underflow (bf16 shares float32's
    exponent range), which may cost relative accuracy only where the PRODUCTS are themselves subnormal; results
    must stay finite and within an absolute 2^-126-scale error plus the usual relative bound."""
    n, h, w, cin, cout = 1, 8, 8, 16, 32
    g = torch.Generator().manual_seed(79)
    x = _wide((n, cin, h, w), g, -100, -80)
    wt = _wide((cout, cin, 3, 3), g, -20, 10)
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    c = ctx()
    dx, dw = c.to_device(nhwc(x)), c.to_device(wt.numpy())
    out = c.empty((n, h, w, cout))
    check(lib.rfi_op_conv3x3(c.handle, impl, P(dx), n, h, w, cin, P(dw), None, cout, None, None, 0, P(out)))
    got = torch.from_numpy(out.numpy()).permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert ((got - want).abs() <= 16 * 2.0 ** -24 * bound + 144 * 2.0 ** -126).all()


# (n, h, w, cin, cout): the stride-2 convolutions of the ResNet-style encoder -- the 2x2 form on the
# space-to-depth input and the 1x1 projection on a channel slice of it, every layer shape of UNetResNet18(f=64) at
# 128 x 128 and of the small test models, plus ragged ones
S2_SHAPES = [(2, 128, 128, 64, 128), (2, 64, 64, 128, 256), (2, 32, 32, 256, 512), (4, 64, 64, 16, 32), (3, 32, 32, 8, 16),
             (1, 8, 8, 4, 8), (1, 12, 20, 20, 36), (3, 4, 4, 32, 64), (2, 16, 16, 64, 64)]


@pytest.mark.parametrize("shape", S2_SHAPES)
@pytest.mark.parametrize("ksize", [3, 1])
def test_stride2_convolutions(shape, ksize):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(11 + hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, ksize, ksize, generator=g) / (ksize * cin ** 0.5)).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=2, padding=ksize // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    c = ctx()
    dxd, dwd, ddy = c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(nhwc(dy))
    for impl in [IMPL_DIRECT, IMPL_MFMA, IMPL_X3]:
        out = c.empty((n, h // 2, w // 2, cout))
        check(lib.rfi_op_conv_s2(c.handle, impl, ksize, P(dxd), n, h, w, cin, P(dwd), cout, P(out)))
        assert rel_err(out.numpy(), nhwc(y.detach())) <= TOL, f"fwd impl={impl}"
        gx = c.empty((n, h, w, cin))
        check(lib.rfi_op_conv_s2_dgrad(c.handle, impl, ksize, P(ddy), n, h, w, cout, P(dwd), cin, P(gx)))
        assert rel_err(gx.numpy(), nhwc(x.grad)) <= TOL, f"dgrad impl={impl}"
        if impl == IMPL_MFMA:
            continue            # these weight gradients exist in the 3 x bf16 / bf16 arithmetic (and the direct kernel) only
        gw = c.empty((cout, cin, ksize, ksize))
        check(lib.rfi_op_conv_s2_wgrad(c.handle, impl, ksize, P(dxd), P(ddy), n, h, w, cin, cout, P(gw)))
        assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, f"wgrad impl={impl}"


IMPL_PBF16 = 6    # plane kernels, bfloat16 flow (the kernels of the bfloat16 compute mode)


@pytest.mark.parametrize("shape", [sh for sh in S2_SHAPES if sh[3] % 4 == 0 and sh[4] % 4 == 0] + [(1, 64, 128, 64, 128), (1, 32, 64, 16, 64)])
@pytest.mark.parametrize("ksize", [3, 1])
def test_stride2_convolutions_on_the_plane_kernels(shape, ksize):
    """The stage transitions of the ResNet-style encoder in the bfloat16 flow: the strided 3x3 / 1x1 contractions read the
    full-resolution bf16 planes (no space-to-depth copy), the input gradient is four 2x2 contractions of dY (one per parity
    class of the input pixel, the 1x1 layer's as a second K segment of class 0), the weight gradient strides its halo tile.
    Against torch on the bf16-ROUNDED operands (products of two bf16 values are exact in float32: the differences are the
    summation order and, for y / dx, the rounding of the stored result to bfloat16)."""
    n, h, w, cin, cout = shape
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    g = torch.Generator().manual_seed(13 + hash(shape) % 1000)
    x = bf(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = bf(torch.randn(cout, cin, ksize, ksize, generator=g) / (ksize * cin ** 0.5)).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=2, padding=ksize // 2)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    c = ctx()
    dxd, dwd, ddy = c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(nhwc(dy))
    out = c.empty((n, h // 2, w // 2, cout))
    check(lib.rfi_op_conv_s2(c.handle, IMPL_PBF16, ksize, P(dxd), n, h, w, cin, P(dwd), cout, P(out)))
    want = nhwc(y.detach())
    assert np.abs(out.numpy() - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6, "fwd"
    assert np.array_equal(out.numpy(), nhwc(bf(nchw(out.numpy())))), "the output holds bf16 values"
    gx = c.empty((n, h, w, cin))
    check(lib.rfi_op_conv_s2_dgrad(c.handle, IMPL_PBF16, ksize, P(ddy), n, h, w, cout, P(dwd), cin, P(gx)))
    want = nhwc(x.grad)
    assert np.abs(gx.numpy() - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6, "dgrad"
    if ksize == 1:
        assert not gx.numpy()[:, 1::2].any() and not gx.numpy()[:, :, 1::2].any()      # pixels no output reads: exact zeros
    gw = c.empty((cout, cin, ksize, ksize))
    check(lib.rfi_op_conv_s2_wgrad(c.handle, IMPL_PBF16, ksize, P(dxd), P(ddy), n, h, w, cin, cout, P(gw)))
    assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, "wgrad"


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 32), (1, 64, 64, 128, 64), (1, 8, 32, 256, 128), (3, 4, 4, 32, 32), (1, 12, 20, 48, 64),
                                   (2, 32, 32, 1024, 512)])
def test_transposed_conv_on_the_plane_kernels(shape):
    """ConvTranspose2d(k2, s2) in the bfloat16 flow: forward = ONE 1x1 contraction on the input planes whose 4 cout output
    channels are the four taps, each written to its own pixel of the 2 x 2 block; input gradient = a 2x2 stride-2
    contraction of dy; weight gradient = the strided pixel reduction with dy as the halo operand.  Against torch on the
    bf16-ROUNDED operands."""
    n, h, w, cin, cout = shape
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    g = torch.Generator().manual_seed(17 + hash(shape) % 1000)
    x = bf(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = bf(torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose2d(x, wt, b, stride=2)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    c = ctx()
    dxd, dwd, dbd, ddy = c.to_device(nhwc(x.detach())), c.to_device(wt.detach().numpy()), c.to_device(b.numpy()), c.to_device(nhwc(dy))
    out = c.empty((n, 2 * h, 2 * w, cout))
    check(lib.rfi_op_convt2x2(c.handle, IMPL_PBF16, P(dxd), n, h, w, cin, P(dwd), P(dbd), cout, P(out)))
    want = nhwc(y.detach())
    assert np.abs(out.numpy() - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6, "fwd"
    gx = c.empty((n, h, w, cin))
    check(lib.rfi_op_convt2x2_dgrad(c.handle, IMPL_PBF16, P(ddy), n, h, w, cout, P(dwd), cin, P(gx)))
    want = nhwc(x.grad)
    assert np.abs(gx.numpy() - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6, "dgrad"
    gw = c.empty((cin, cout, 2, 2))
    check(lib.rfi_op_convt2x2_wgrad(c.handle, IMPL_PBF16, P(dxd), P(ddy), n, h, w, cin, cout, P(gw)))
    assert rel_err(gw.numpy(), wt.grad.numpy()) <= 5e-5, "wgrad"


@pytest.mark.parametrize("shape", [(1, 8, 32, 64, 64), (2, 4, 4, 64, 64), (1, 8, 32, 3136, 128), (250, 1, 1, 3136, 128), (1, 8, 32, 256, 1024),
                                   (1, 8, 32, 2048, 512), (1, 8, 32, 64, 32), (3, 5, 7, 32, 96)])
@pytest.mark.parametrize("xform", [False, True])
def test_conv1x1(shape, xform):
    """1x1 stride-1 conv = a GEMM over the pixels (the Bottleneck / pyramid / fully connected layers of the detector): the
    round-2 kernel (auto) and the wave-specialised GEMM kernel (IMPL_WS) against torch."""
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    xin = torch.relu(x * sc + sh) if xform else x
    want = (xin.reshape(-1, cin) @ wt.reshape(cout, cin).T + b).reshape(n, h, w, cout).numpy()
    c = ctx()
    dx, dw, db = c.to_device(x.numpy()), c.to_device(wt.numpy()), c.to_device(b.numpy())
    dsc, dsh = c.to_device(sc.numpy()), c.to_device(sh.numpy())
    for impl in (0, IMPL_WS):
        dy = c.empty((n, h, w, cout))
        check(lib.rfi_op_conv1x1(c.handle, impl, P(dx), n, h, w, cin, P(dw), P(db), cout, P(dsc) if xform else None,
                                 P(dsh) if xform else None, 1 if xform else 0, P(dy)))
        assert rel_err(dy.numpy(), want) <= TOL, f"impl={impl}"
