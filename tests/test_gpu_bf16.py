"""-m gpu: the bfloat16 compute mode (MFMA operands rounded to bf16 in registers, float32 accumulate and
storage) -- a builder-chosen reduced-precision mode (the reference's CPU path is float32; on a
GPU it autocasts to float16 with a GradScaler, scripts/train_model.py:131,144).  Not bit-comparable with the float32 CPU reference: checked against
an oracle that rounds the same operands (weights, activations, output gradients) to bf16, and by the
north-star criterion, |dIoU| <= 1e-3 on the reference-trained weights."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import metrics_ref, unet_ref
from rfi_toolbox_amd.models import UNet

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def test_bf16_conv_rounds_operands_only():
    """One 3x3 layer through the C ABI: bf16 mode == float32 conv of bf16-rounded input and weights
    (products of two bf16 values are exact in float32; only the summation order differs)."""
    import ctypes as C

    from rfi_toolbox_amd._lib import check, lib
    from rfi_toolbox_amd.runtime import Context
    torch.manual_seed(3)
    m = UNet(3, 1, 16).set_compute_dtype("bfloat16").eval()
    m32 = UNet(3, 1, 16).load_state_dict(m.state_dict()).eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 32, 32, generator=g)
    a, b = m(x).numpy(), m32(x).numpy()
    rel = np.abs(a - b).max() / np.abs(b).max()
    assert 1e-5 < rel < 3e-2, rel                      # differs from float32 (it IS bf16), by a bf16-sized amount
    assert m.set_compute_dtype("float32")(x).numpy().tobytes() == b.tobytes()   # and the switch goes back


def test_bf16_inference_iou_on_reference_weights(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_f8_b4_s64.npz"))
    st = OrderedDict((k[8:], torch.from_numpy(g[k].copy())) for k in g.files if k.startswith("state40/"))
    m = UNet(3, 1, 8).load_state_dict(st).eval().set_compute_dtype("bfloat16")
    logits = m.forward_nhwc(g["img"])[..., 0]
    want = g["logits_eval40"][:, 0]
    lab = g["lab"]
    got_m, ref_m = metrics_ref.evaluate_segmentation(logits > 0, lab), metrics_ref.evaluate_segmentation(want > 0, lab)
    for k in ("iou", "precision", "recall", "f1", "dice"):
        assert abs(got_m[k] - ref_m[k]) <= 1e-3, (k, got_m[k], ref_m[k])
    assert np.abs(logits - want).max() <= 0.05 * np.abs(want).max()


def _golden_state(npz, tag):
    return OrderedDict((k[len(tag) + 1:], torch.from_numpy(npz[k].copy())) for k in npz.files if k.startswith(tag + "/"))


def _oracle_bf16_trajectory(g, steps, lr, wd, iou_every=0, **adam_kw):
    """The oracle in the bf16-operand arithmetic on the golden inputs: what a CORRECT bf16 implementation reaches, i.e.
    the calibration of how far the arithmetic itself moves the reference's float32 trajectory."""
    st = _golden_state(g, "state0")
    adam = unet_ref.new_adam_state(st)
    x, y = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"])), torch.from_numpy(g["lab"]).float().unsqueeze(1)
    losses, ious, states = [], {}, {}
    for s in range(1, steps + 1):
        with unet_ref.bf16_operands(round_outputs=True):
            losses.append(unet_ref.train_step(st, adam, x, y, lr=lr, weight_decay=wd, **adam_kw)["loss"])
            if iou_every and s % iou_every == 0:
                with torch.no_grad():
                    lg = unet_ref.forward(st, x, training=False)
                ious[s] = metrics_ref.evaluate_segmentation(lg[:, 0].numpy() > 0, g["lab"])["iou"]
        states[s] = OrderedDict((k, v.clone()) for k, v in st.items())
    return losses, ious, states, adam


def test_bf16_trajectory_against_the_reference_golden_f8(golden_dir):
    """The REFERENCE's 40-step training trajectory (tests/golden/unet_f8_b4_s64.npz, captured from
    /root/reference/rfi_toolbox/scripts/train_model.py:139-151 semantics on CPU float32) replayed in the bfloat16 mode.
    Bounds: training is a chaotic map of its rounding errors, and bf16 operands perturb every contraction by ~2^-9, so
    the yardstick is the CPU oracle run in the same bf16-operand arithmetic on the same data (it leaves the golden IoU
    by up to 4e-3 and the golden loss by 7e-4): the HIP trajectory may sit twice as far from the golden values as that
    oracle does (floors 3e-3 IoU / 1e-3 loss) and must stay within 4e-3 IoU / 2e-3 loss of the oracle itself."""
    g = np.load(os.path.join(golden_dir, "unet_f8_b4_s64.npz"))
    o_loss, o_iou, _, _ = _oracle_bf16_trajectory(g, 40, 1e-3, 1e-5, iou_every=10)
    m = UNet(3, 1, 8).load_state_dict(_golden_state(g, "state0")).set_compute_dtype("bfloat16")
    gold_loss = [float(v) for v in g["losses"]]
    o_dev = max(abs(a - b) for a, b in zip(o_loss, gold_loss))
    ious = {}
    for s in range(1, 41):
        loss = m.train_step(g["img"], g["lab"], lr=1e-3, weight_decay=1e-5)
        assert abs(loss - gold_loss[s - 1]) <= max(2 * o_dev, 1e-3), (s, loss, gold_loss[s - 1], o_dev)
        assert abs(loss - o_loss[s - 1]) <= 2e-3, (s, loss, o_loss[s - 1])
        if s % 10 == 0:
            m.eval()
            ious[s] = metrics_ref.evaluate_segmentation(m.forward_nhwc(g["img"])[..., 0] > 0, g["lab"])["iou"]
            m.train()
    for s, want in zip(g["iou_steps"], g["iou"]):
        s, want = int(s), float(want)
        assert abs(ious[s] - want) <= max(2 * abs(o_iou[s] - want), 3e-3), (s, ious[s], want, o_iou[s])
        assert abs(ious[s] - o_iou[s]) <= 4e-3, (s, ious[s], o_iou[s])
    assert ious[40] >= float(g["iou"][-1]) - 3e-3          # it trains to the reference's final IoU


def test_bf16_three_steps_against_the_reference_golden_f4(golden_dir):
    """The reference's first three optimisation steps (tests/golden/unet_f4_b4_s32.npz) in the bfloat16 mode: losses,
    every parameter after steps 1 and 3, Adam moments after step 3.  Adam normalises an update to ~lr whatever the
    gradient's precision, so a parameter can differ from the float32 golden value by at most ~2 lr per step (a sign
    flip of a noise-level gradient) and typically by a few per cent of lr; the moments carry the gradient's bf16 error
    directly and are bounded by twice what the bf16-operand oracle itself shows against the golden values."""
    g = np.load(os.path.join(golden_dir, "unet_f4_b4_s32.npz"))
    lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
    o_loss, _, o_states, o_adam = _oracle_bf16_trajectory(g, 3, lr, wd, betas=(b1, b2), eps=eps, clip=clip)
    m = UNet(3, 1, 4).load_state_dict(_golden_state(g, "state0")).set_compute_dtype("bfloat16")
    for s in (1, 2, 3):
        loss = m.train_step(g["img"], g["lab"], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, max_grad_norm=clip)
        assert abs(loss - float(g["losses"][s - 1])) <= 1e-3 and abs(loss - o_loss[s - 1]) <= 1e-3, (s, loss)
        if s in (1, 3):
            sd = m.state_dict()
            meds = []
            for k, v in sd.items():
                want = g[f"state{s}/{k}"]
                if k.endswith("num_batches_tracked"):
                    assert int(v) == int(want), k
                elif "running_" in k:        # BatchNorm buffers follow the bf16 activations: against the same-arithmetic oracle
                    np.testing.assert_allclose(v.numpy(), o_states[s][k].numpy(), rtol=2e-2, atol=2e-3, err_msg=k)
                else:
                    d = np.abs(v.numpy() - want)
                    assert d.max() <= 2.2 * lr * s, (k, d.max() / (lr * s))
                    meds.append(np.median(d) / (lr * s))
            assert np.median(meds) <= 0.25, np.median(meds)        # the typical parameter: a fraction of one update
    for k in ("encoder1.conv.conv.0.weight", "decoder2.up.weight", "final_conv.weight"):
        mm, vv, step = m.adam_state(k)
        assert step == 3
        wm, wv = g[f"adam_m3/{k}"], g[f"adam_v3/{k}"]
        om, ov = o_adam["m"][k].numpy(), o_adam["v"][k].numpy()
        rel = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)   # noqa: E731
        assert rel(mm, wm) <= max(2 * rel(om, wm), 2e-2), (k, rel(mm, wm), rel(om, wm))
        assert rel(vv, wv) <= max(2 * rel(ov, wv), 4e-2), (k, rel(vv, wv), rel(ov, wv))


def test_bf16_unet_1024_vs_bf16_operand_oracle():
    """BASELINE configs[2] shape in its stated arithmetic (SURVEY 8a A10 stand-in): UNet(3,1,32), one
    1024x1024x3 waterfall, bf16 compute mode, forward + loss + backward against the oracle run in the SAME
    arithmetic (`unet_ref.bf16_operands()`: every contraction sees bf16-rounded operands, float32 accumulate).
    What is left between the two is summation order and the handful of activations that round to the other
    side of a bf16 tie or the ReLU threshold."""
    st = unet_ref.init_state(3, 1, 32, seed=3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 1024, 1024, 3, generator=g)
    y = (torch.rand(1, 1024, 1024, generator=g) > 0.9).to(torch.uint8)
    y[:, 300:340, :] = 1
    xo, yo = unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)
    with unet_ref.bf16_operands(round_outputs=True):
        lbf, lgbf, gbf, _ = unet_ref.loss_and_grads(st, xo, yo)
    l32, lg32, g32, _ = unet_ref.loss_and_grads(st, xo, yo)
    m = UNet(3, 1, 32).load_state_dict(st).train().set_compute_dtype("bfloat16")
    loss = m.forward_backward(x, y)
    got = m.debug_tensor("logits")
    want = lgbf.permute(0, 2, 3, 1).reshape(-1).numpy()
    span = float(np.abs(want).max())
    d_same = np.abs(got - want).max()                    # against the oracle in the same arithmetic
    d_f32 = np.abs(lg32.permute(0, 2, 3, 1).reshape(-1).numpy() - want).max()   # what the arithmetic itself costs
    assert d_same <= 0.02 * span and d_same <= max(0.5 * d_f32, 2e-3 * span), (d_same, d_f32, span)
    assert loss == pytest.approx(float(lbf), rel=2e-3)
    for k in ("final_conv.weight", "decoder1.conv.conv.3.weight", "decoder1.up.weight", "decoder3.conv.conv.0.weight",
              "bottleneck.conv.3.weight", "encoder3.conv.conv.0.weight", "encoder1.conv.conv.3.weight",
              "encoder1.conv.conv.0.weight", "encoder2.conv.conv.1.weight", "decoder2.conv.conv.4.bias"):
        w_bf, w_32 = gbf[k].numpy().ravel(), g32[k].numpy().ravel()
        nrm = np.linalg.norm(w_bf) + 1e-30
        rel_same = np.linalg.norm(m.grad(k).ravel() - w_bf) / nrm
        rel_arith = np.linalg.norm(w_32 - w_bf) / nrm
        # closer to the same-arithmetic oracle than the float32 oracle is (see test_gpu_bench_config.py)
        assert rel_same <= max(1.0 * rel_arith, 1e-2), (k, rel_same, rel_arith)


def test_bf16_inference_iou_at_1024_on_reference_weights(golden_dir):
    """|dIoU| <= 1e-3 (north_star) at the configs[2] size: the weights the REFERENCE trained (tests/golden,
    f = 8) applied to a 1024x1024 synthetic waterfall from the device generator + Preprocessor pipeline,
    bf16 mode on the GPU against the float32 oracle, both scored against the generator's own RFI mask."""
    from rfi_toolbox_amd.data_generation import make_training_patches_device
    g = np.load(os.path.join(golden_dir, "unet_f8_b4_s64.npz"))
    st = OrderedDict((k[8:], torch.from_numpy(g[k].copy())) for k in g.files if k.startswith("state40/"))
    d_x, d_y = make_training_patches_device(1, 1024, seed=77, device=0)
    x, lab = d_x.numpy(), d_y.numpy()
    assert x.shape == (1, 1024, 1024, 3) and 0.005 < lab.mean() < 0.9
    m = UNet(3, 1, 8).load_state_dict(st).eval().set_compute_dtype("bfloat16")
    logits = m.forward_nhwc(x)[..., 0]
    with torch.no_grad():
        want = unet_ref.forward(st, unet_ref.nhwc_to_nchw(torch.from_numpy(x)), training=False)[:, 0].numpy()
    got_m, ref_m = metrics_ref.evaluate_segmentation(logits > 0, lab), metrics_ref.evaluate_segmentation(want > 0, lab)
    for k in ("iou", "precision", "recall", "f1", "dice"):
        assert abs(got_m[k] - ref_m[k]) <= 1e-3, (k, got_m[k], ref_m[k])
    assert ((logits > 0) != (want > 0)).mean() <= 1e-3


@pytest.mark.parametrize("cls,f,n,h,w", [("UNetBigger", 8, 2, 64, 64), ("UNet", 12, 2, 48, 80), ("UNet", 6, 1, 32, 32),
                                         ("UNet", 16, 3, 16, 16), ("UNet", 16, 2, 48, 80)])
def test_bf16_data_flow_other_shapes(cls, f, n, h, w):
    """The bf16 data flow away from the benched shape: depth 5 (1024 channels at the bottleneck), channel counts
    that are not powers of two (partial 16-channel chunks, the scalar fallbacks of the fused reductions), a width
    that is not a multiple of 4 (conv outputs stay float32 there: the oracle then rounds operands only), tiny maps, and
    widths of 16 (bfloat16 gradient tensors, BatchNorm-backward sums in the conv epilogue) on maps that end in partial
    tiles."""
    from rfi_toolbox_amd.models import UNetBigger
    depth = 5 if cls == "UNetBigger" else 4
    st = unet_ref.init_state(3, 1, f, depth=depth, seed=7)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, h, w, 3, generator=g)
    y = (torch.rand(n, h, w, generator=g) > 0.8).to(torch.uint8)
    xo, yo = unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)
    with unet_ref.bf16_operands(round_outputs=(f % 4 == 0)):
        lb, lgb, gb, _ = unet_ref.loss_and_grads(st, xo, yo)
    l32, lg32, g32, _ = unet_ref.loss_and_grads(st, xo, yo)
    m = (UNetBigger if cls == "UNetBigger" else UNet)(3, 1, f).load_state_dict(st).train().set_compute_dtype("bfloat16")
    loss = m.forward_backward(x, y)
    assert loss == pytest.approx(float(lb), rel=5e-3)
    want = lgb.permute(0, 2, 3, 1).reshape(-1).numpy()
    w32 = lg32.permute(0, 2, 3, 1).reshape(-1).numpy()
    span = float(np.abs(want).max())
    e_same, e_arith = np.abs(m.debug_tensor("logits") - want), np.abs(w32 - want)
    d_same, d_arith = e_same.max(), e_arith.max()
    # closer to the oracle in the SAME arithmetic than that oracle is to float32: the typical logit by a factor 0.6, the
    # worst one by 0.75 (a single bf16 rounding flip of a conv output moves a logit by about half of d_arith; measured
    # max 0.38-0.61 of d_arith over seeds at 48 x 80, median 0.33-0.49)
    assert d_same <= max(0.75 * d_arith, 5e-3 * span), (d_same, d_arith, span)
    assert np.median(e_same) <= max(0.6 * np.median(e_arith), 1e-3 * span), (np.median(e_same), np.median(e_arith))
    rels = []
    for k, gk in gb.items():
        if k.endswith((".0.bias", ".3.bias")) and "conv" in k:
            continue
        gk = gk.numpy().ravel()
        nrm = np.linalg.norm(gk) + 1e-30
        rel_same = np.linalg.norm(m.grad(k).ravel() - gk) / nrm
        rel_arith = np.linalg.norm(g32[k].numpy().ravel() - gk) / nrm
        rels.append(rel_same / max(rel_arith, 1e-9))
        assert rel_same <= max(1.5 * rel_arith, 3e-2), (k, rel_same, rel_arith)
    assert np.median(rels) <= 1.0, np.median(rels)
