"""-m gpu: the per-RoI mask branch of Mask R-CNN (BASELINE.json configs[3]) on MI355X against
oracle/mask_head_ref.py (the same layers as plain torch.nn modules; parity unpinned by the reference, which has no
detector), and the whole branch -- RoIAlign -> mask head -> BCE -> gradients back into the feature map -- against the
NumPy RoIAlign oracle chained with the torch head."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import detection_ref, mask_head_ref as mref, unet_ref
from rfi_toolbox_amd.models import MaskHead
from rfi_toolbox_amd.models import detection_ops as ops

pytestmark = pytest.mark.gpu


def _case(c, r, s, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(r, s, s, c, generator=g)
    y = (torch.rand(r, 2 * s, 2 * s, generator=g) > 0.6).to(torch.uint8)
    return x, y, unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)


def test_state_dict_init_and_forward():
    torch.manual_seed(5)
    m = MaskHead(16, 1, 4)
    want = mref.init_state(16, 1, 4, seed=5)
    sd = m.state_dict()
    assert list(sd.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(sd[k], want[k]), k
    x, _, xo, _ = _case(16, 5, 14, 6)
    got = m.eval()(xo)
    ref = mref.forward(want, xo)
    assert got.shape == (5, 1, 28, 28)
    np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()) + 1e-6)
    np.testing.assert_allclose(m.forward_nhwc(x.numpy())[..., 0], ref.detach().numpy()[:, 0], rtol=0,
                               atol=2e-5 * float(ref.abs().max()) + 1e-6)
    with pytest.raises(ValueError):
        MaskHead(10, 1)
    with pytest.raises(ValueError):
        m.train_step(x, torch.zeros(5, 14, 14, dtype=torch.uint8))       # labels live on the 2x map


@pytest.mark.parametrize("mode,c,r,s", [("float32", 16, 6, 14), ("float32_mfma", 16, 6, 14), ("float32", 64, 32, 14), ("float32", 256, 8, 14)])
def test_gradients_vs_oracle(mode, c, r, s):
    st = mref.init_state(c, 1, 4, seed=11)
    x, y, xo, yo = _case(c, r, s, 12)
    l32, lg32, g32, gx32 = mref.loss_and_grads(st, xo, yo)
    st64 = OrderedDict((k, v.double()) for k, v in st.items())
    _, _, g64, gx64 = mref.loss_and_grads(st64, xo.double(), yo.double())
    m = MaskHead(c, 1, 4).load_state_dict(st).train().set_compute_dtype(mode)
    loss = m.forward_backward(x, y)
    assert loss == pytest.approx(float(l32), abs=5e-6)
    want = lg32.permute(0, 2, 3, 1).reshape(-1).numpy()
    assert np.abs(m.debug_tensor("logits") - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max()))
    got_all = {k: m.grad(k) for k in g64}
    got_all["input"] = m.input_grad((r, s, s, c))
    g64 = dict(g64, input=gx64.permute(0, 2, 3, 1))
    g32 = dict(g32, input=gx32.permute(0, 2, 3, 1))
    for k, w64 in g64.items():
        w64 = w64.numpy().ravel()
        nrm = np.linalg.norm(w64) + 1e-30
        rel_ref = np.linalg.norm(g32[k].numpy().ravel() - w64) / nrm
        rel_hip = np.linalg.norm(got_all[k].ravel() - w64) / nrm
        # (plain ReLUs on conv outputs, no BatchNorm: threshold flips are rare; 1e-3 covers one)
        assert rel_hip <= max(4 * rel_ref, 1e-3), (k, rel_hip, rel_ref)


def test_three_training_steps_vs_oracle():
    c, r, s = 16, 8, 14
    st = mref.init_state(c, 1, 4, seed=21)
    x, y, xo, yo = _case(c, r, s, 22)
    m = MaskHead(c, 1, 4).load_state_dict(st).train()
    ost = OrderedDict((k, v.clone()) for k, v in st.items())
    adam = unet_ref.new_adam_state(ost)
    for step in range(3):
        ref = mref.train_step(ost, adam, xo, yo, lr=1e-3, weight_decay=1e-5, clip=1.0)
        loss = m.train_step(x, y, lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        assert loss == pytest.approx(ref["loss"], abs=2e-5 * (step + 1)), step
    sd = m.state_dict()
    for k, v in ost.items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=0, atol=5e-4, err_msg=k)
    tp, fp, fn = m.eval_batch(x, y)
    pred = (mref.forward(ost, xo) > 0).numpy()[:, 0]
    yt = y.numpy() > 0
    assert abs(tp - int((pred & yt).sum())) <= 3 and abs(fp - int((pred & ~yt).sum())) <= 3 and abs(fn - int((~pred & yt).sum())) <= 3


def test_bf16_operand_mode():
    c, r, s = 32, 16, 14
    st = mref.init_state(c, 1, 4, seed=31)
    x, y, xo, yo = _case(c, r, s, 32)
    l32, _, g32, _ = mref.loss_and_grads(st, xo, yo)
    with unet_ref.bf16_operands():
        lb, _, gb, gxb = mref.loss_and_grads(st, xo, yo)
    m = MaskHead(c, 1, 4).load_state_dict(st).train().set_compute_dtype("bfloat16")
    loss = m.forward_backward(x, y)
    assert loss == pytest.approx(float(lb), rel=2e-3)
    for k, g in gb.items():
        g = g.numpy().ravel()
        nrm = np.linalg.norm(g) + 1e-30
        rel_same = np.linalg.norm(m.grad(k).ravel() - g) / nrm
        rel_arith = np.linalg.norm(g32[k].numpy().ravel() - g) / nrm
        assert rel_same <= max(1.1 * rel_arith, 1e-2), (k, rel_same, rel_arith)


def test_mask_branch_end_to_end():
    """features -> RoIAlign (14 x 14, sampling 2) -> mask head -> BCE against per-RoI targets -> gradient of the
    feature map, every stage on the GPU, against the NumPy RoIAlign oracle chained with the torch head."""
    rng = np.random.default_rng(3)
    n, size, c, scale = 2, 32, 16, 0.25
    feats = rng.standard_normal((n, size, size, c)).astype(np.float32)
    rois = np.array([[0, 4, 4, 60, 70], [1, 20, 10, 90, 100], [0, 0, 0, 128, 128], [1, 50.5, 40.25, 75, 61], [0, 100, 90, 140, 131],
                     [1, 8, 8, 24, 24]], dtype=np.float32)
    r = len(rois)
    target = (rng.random((r, 28, 28)) > 0.5).astype(np.uint8)
    st = mref.init_state(c, 1, 4, seed=41)
    # oracle chain
    roi_feats = detection_ref.roi_align(feats, rois, scale, (14, 14), 2, False)
    xo = torch.from_numpy(np.ascontiguousarray(roi_feats.transpose(0, 3, 1, 2))).float()
    l_ref, _, g_ref, gx_ref = mref.loss_and_grads(st, xo, torch.from_numpy(target).float().unsqueeze(1))
    dfeat_ref = detection_ref.roi_align_backward(gx_ref.permute(0, 2, 3, 1).numpy().astype(np.float64), feats.shape, rois, scale,
                                                 (14, 14), 2, False)
    # HIP chain
    m = MaskHead(c, 1, 4).load_state_dict(st).train()
    rf = ops.roi_align(feats, rois, scale, (14, 14), 2, False)
    loss = m.forward_backward(rf, target)
    dfeat = ops.roi_align_backward(m.input_grad(rf.shape), feats.shape, rois, scale, 2, False)
    assert loss == pytest.approx(float(l_ref), abs=1e-5)
    assert np.abs(dfeat - dfeat_ref).max() <= 2e-4 * np.abs(dfeat_ref).max()
    k = "mask_fcn1.weight"
    assert np.linalg.norm(m.grad(k) - g_ref[k].numpy()) <= 2e-4 * np.linalg.norm(g_ref[k].numpy())


# ---------------------------------------------------------------- RPN head
def test_rpn_head_step_and_proposals():
    """RPN head: forward, the loss kernel, backward from its gradient, one Adam step; then proposals from the head output
    (box decoding + NMS) -- against the torch head / torch loss / NumPy decode + NMS oracles."""
    from rfi_toolbox_amd.models import RPNHead
    c, a, n, hh, ww = 32, 4, 2, 24, 20
    st = mref.rpn_init_state(c, a, 1, seed=51)
    torch.manual_seed(51)
    m = RPNHead(c, a, 1)
    sd = m.state_dict()
    assert list(sd.keys()) == list(st.keys())
    for k in st:
        assert torch.equal(sd[k], st[k]), k                              # same draws as constructing the torch module
    rng = np.random.default_rng(52)
    x = torch.from_numpy(rng.standard_normal((n, hh, ww, c)).astype(np.float32))
    xo = unet_ref.nhwc_to_nchw(x)
    want = mref.rpn_forward(st, xo)
    got = m.train().forward_nhwc(x.numpy())
    assert got.shape == (n, hh, ww, 5 * a)
    np.testing.assert_allclose(got, want.detach().numpy(), rtol=0, atol=3e-5 * float(want.abs().max()))
    # sampler output (synthetic): ~3 % positives, ~6 % negatives, the rest not sampled
    P = n * hh * ww
    labels = rng.choice(np.array([-1] * 30 + [0, 0, 1], np.int8), P * a)
    targets = (rng.standard_normal((P * a, 4)) * 0.3).astype(np.float32)
    lo, lb, dout = ops.rpn_loss(got.reshape(P, 5 * a), labels, targets, a)
    # oracle: torch autograd through head + loss
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in st.items())
    xg = xo.clone().requires_grad_(True)
    o_l, b_l = mref.rpn_loss_torch(mref.rpn_forward(leaves, xg), labels, targets, a)
    grads = torch.autograd.grad(o_l + b_l, list(leaves.values()) + [xg])
    assert lo == pytest.approx(float(o_l), rel=1e-5) and lb == pytest.approx(float(b_l), rel=1e-5)
    m.backward(x.numpy(), dout)
    for (k, _), g in zip(leaves.items(), grads[:-1]):
        g = g.numpy()
        assert np.linalg.norm(m.grad(k) - g) <= 1e-4 * np.linalg.norm(g) + 1e-9, k
    gx = grads[-1].permute(0, 2, 3, 1).numpy()
    assert np.linalg.norm(m.input_grad((n, hh, ww, c)) - gx) <= 1e-4 * np.linalg.norm(gx)
    norm = m.apply_gradients(lr=1e-3, weight_decay=0.0)
    total, _ = unet_ref.clip_coefficient(OrderedDict(zip(leaves.keys(), grads[:-1])), 1.0)
    assert norm == pytest.approx(float(total), rel=1e-4)
    # proposals of image 0: anchors of 4 sizes on a stride-8 grid, top 600 by objectness, NMS at 0.7
    sizes = np.array([16, 32, 64, 128], np.float32)
    ys, xs = np.meshgrid(np.arange(hh) * 8 + 4, np.arange(ww) * 8 + 4, indexing="ij")
    anchors = np.stack([(xs[..., None] - sizes / 2), (ys[..., None] - sizes / 2), (xs[..., None] + sizes / 2), (ys[..., None] + sizes / 2)],
                       -1).reshape(-1, 4).astype(np.float32)             # (H W A, 4), pixel-major, anchor-minor
    out0 = got[0].reshape(hh * ww, 5 * a)
    scores, deltas = out0[:, :a].reshape(-1), out0[:, a:].reshape(-1, 4)
    boxes = ops.decode_boxes(anchors, deltas, image_size=(hh * 8, ww * 8))
    assert np.abs(boxes - detection_ref.decode_boxes(anchors, deltas, image_size=(hh * 8, ww * 8))).max() <= 1e-3
    top = np.argsort(-scores, kind="stable")[:600]
    keep = ops.nms(boxes[top], scores[top], 0.7)
    np.testing.assert_array_equal(keep, detection_ref.nms(boxes[top], scores[top], 0.7))
    assert 0 < len(keep) <= 600


def test_rpn_head_shared_over_two_levels_accumulates_gradients():
    """One RPN head applied to two pyramid levels: the parameter gradients of the two backward passes are summed by
    `accumulate_gradients` and equal the oracle's gradient of the sum of the two losses."""
    from rfi_toolbox_amd.models import RPNHead
    c, a, n = 16, 4, 2
    st = mref.rpn_init_state(c, a, 1, seed=61)
    m = RPNHead(c, a, 1).load_state_dict(st).train()
    rng = np.random.default_rng(62)
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in st.items())
    total = 0
    m.accumulate_gradients("begin")
    for hh in (16, 8):
        x = rng.standard_normal((n, hh, hh, c)).astype(np.float32)
        P = n * hh * hh
        labels = rng.choice(np.array([-1] * 8 + [0, 0, 1], np.int8), P * a)
        targets = (rng.standard_normal((P * a, 4)) * 0.3).astype(np.float32)
        out = m.forward_nhwc(x)
        _, _, dout = ops.rpn_loss(out.reshape(P, 5 * a), labels, targets, a)
        m.backward(x, dout)
        m.accumulate_gradients("add")
        o_l, b_l = mref.rpn_loss_torch(mref.rpn_forward(leaves, unet_ref.nhwc_to_nchw(torch.from_numpy(x))), labels, targets, a)
        total = total + o_l + b_l
    m.accumulate_gradients("end")
    grads = torch.autograd.grad(total, list(leaves.values()))
    for (k, _), g in zip(leaves.items(), grads):
        g = g.numpy()
        assert np.linalg.norm(m.grad(k) - g) <= 1e-4 * np.linalg.norm(g) + 1e-9, k


# ---------------------------------------------------------------- box head
@pytest.mark.parametrize("c,res,hid,k1,r", [(16, 7, 64, 2, 96), (256, 7, 1024, 2, 512), (8, 3, 32, 5, 37)])
def test_box_head_step_vs_oracle(c, res, hid, k1, r):
    """TwoMLPHead + FastRCNNPredictor: default init == constructing the torch modules, forward, the Fast R-CNN loss kernel,
    backward from its gradient (every parameter, the RoI features), one Adam step -- against torch autograd."""
    from rfi_toolbox_amd.models import BoxHead
    torch.manual_seed(71)
    mod = mref.BoxHeadModule(c, res, hid, k1)
    torch.manual_seed(71)
    m = BoxHead(c, res, hid, k1).train()
    sd = m.state_dict()
    assert list(sd.keys()) == list(mod.state_dict().keys())
    for k, v in mod.state_dict().items():
        assert torch.equal(sd[k], v), k
    rng = np.random.default_rng(72)
    x = rng.standard_normal((r, res, res, c)).astype(np.float32)                 # NHWC, as roi_align returns it
    xo = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    cls, box = mod(xo)
    got = m.forward_rois(x)
    want = torch.cat([cls, box], 1).detach().numpy()
    assert np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max())
    labels = rng.integers(0, k1, r).astype(np.int32)
    targets = (rng.standard_normal((r, 4)) * 0.3).astype(np.float32)
    lc, lb, dout = ops.fastrcnn_loss(got, labels, targets)
    o_c, o_b = mref.fastrcnn_loss_torch(cls, box, labels, targets)
    assert lc == pytest.approx(float(o_c.detach()), rel=1e-5) and lb == pytest.approx(float(o_b.detach()), rel=1e-5, abs=1e-7)
    params = list(mod.parameters())
    grads = torch.autograd.grad(o_c + o_b, params + [xo])
    m.backward(x, dout)
    for (k, _), g in zip(mod.named_parameters(), grads[:-1]):
        g = g.numpy()
        assert np.linalg.norm(m.grad(k) - g) <= 2e-4 * np.linalg.norm(g) + 1e-9, k
    gx = grads[-1].permute(0, 2, 3, 1).numpy()
    assert np.linalg.norm(m.input_grad(x.shape) - gx) <= 2e-4 * np.linalg.norm(gx)
    norm = m.apply_gradients(lr=1e-3, weight_decay=0.0)
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads[:-1])))
    assert norm == pytest.approx(total, rel=2e-4)
