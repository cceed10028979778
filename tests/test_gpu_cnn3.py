"""-m gpu: the "3-layer CNN segmenter" (BASELINE.json configs[0]/[1]) on
MI355X against its golden trajectory and the CPU oracle at the configs[1] width (64)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import cnn_ref, unet_ref
from rfi_toolbox_amd.models import SimpleCNN

pytestmark = pytest.mark.gpu


def _state(npz, tag):
    return OrderedDict((k[len(tag) + 1:], torch.from_numpy(npz[k].copy())) for k in npz.files
                       if k.startswith(tag + "/"))


def test_cnn3_state_dict_and_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn3_c16_b4_s32.npz"))
    m = SimpleCNN(3, 1, 16)
    st = _state(g, "state0")
    m.load_state_dict(st)
    back = m.state_dict()
    assert list(back.keys()) == [str(n) for n in g["names"]]
    for k, v in st.items():
        assert torch.equal(back[k], v), k
    assert m.num_parameters() == sum(v.numel() for v in st.values())
    x_nchw = torch.from_numpy(g["img"]).permute(0, 3, 1, 2).contiguous()
    np.testing.assert_allclose(m.eval()(x_nchw).numpy(), g["logits_eval0"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.forward_nhwc(g["img"])[..., 0], g["logits_eval0"][:, 0], rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2, 32, 32))
    with pytest.raises(RuntimeError):
        m.load_state_dict({"encoder.0.weight": torch.zeros(1)})


def test_cnn3_three_steps_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn3_c16_b4_s32.npz"))
    assert g["relu_margin"].min() > 5e-6
    lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
    m = SimpleCNN(3, 1, 16).load_state_dict(_state(g, "state0")).train()
    for s in (1, 2, 3):
        loss = m.forward_backward(g["img"], g["lab"])
        if s == 1:
            gn = float(g["grad_norms"][0])
            coef = min(1.0, clip / (gn + 1e-6))
            for k in [k[6:] for k in g.files if k.startswith("grad1/")]:
                want = g[f"grad1/{k}"]
                rel = np.linalg.norm(m.grad(k) * coef - want) / (np.linalg.norm(want) + 1e-30)
                assert rel <= 2e-5, (k, rel)
        norm = m.apply_gradients(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, max_grad_norm=clip)
        assert norm == pytest.approx(float(g["grad_norms"][s - 1]), rel=1e-5)
        assert loss == pytest.approx(float(g["losses"][s - 1]), abs=2e-6)
        if s in (1, 3):
            for k, v in m.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g[f"state{s}/{k}"], rtol=0, atol=2e-5, err_msg=k)
            ev = m.eval().forward_nhwc(g["img"])
            m.train()
            np.testing.assert_allclose(ev[..., 0], g[f"logits_eval{s}"][:, 0], rtol=0, atol=1e-4)
    for k in ("encoder.0.weight", "encoder.2.weight", "decoder.0.bias"):
        mm, vv, step = m.adam_state(k)
        assert step == 3
        wm, wv = g[f"adam_m3/{k}"], g[f"adam_v3/{k}"]
        assert np.linalg.norm(mm - wm) <= 2e-5 * np.linalg.norm(wm), k
        assert np.linalg.norm(vv - wv) <= 2e-5 * np.linalg.norm(wv), k


def test_cnn3_width64_vs_oracle():
    """configs[1] width on seeded inputs at 8 x 64 x 64: loss, logits and every gradient tensor
    against oracle/cnn_ref.py; the bound is calibrated by the same oracle run in float64."""
    st = cnn_ref.init_state(3, 1, 64, seed=11)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(8, 64, 64, 3, generator=g)
    y = (torch.rand(8, 64, 64, generator=g) > 0.85).to(torch.uint8)
    xo, yo = unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)
    l32, lg32, g32, _ = cnn_ref.loss_and_grads(st, xo, yo)
    st64 = OrderedDict((k, v.double()) for k, v in st.items())
    _, _, g64, _ = unet_ref.loss_and_grads(st64, xo.double(), yo.double(), forward_fn=cnn_ref.forward)
    m = SimpleCNN(3, 1, 64).load_state_dict(st).train()
    loss = m.forward_backward(x, y)
    assert loss == pytest.approx(float(l32), abs=2e-6)
    np.testing.assert_allclose(m.debug_tensor("logits"), lg32.permute(0, 2, 3, 1).reshape(-1).numpy(),
                               rtol=0, atol=5e-6)
    for k in g32:
        ref64 = g64[k].numpy()
        e_ref = np.linalg.norm(g32[k].numpy() - ref64)
        e_hip = np.linalg.norm(m.grad(k) - ref64)
        assert e_hip <= 4 * e_ref + 1e-6 * np.linalg.norm(ref64), (k, e_hip, e_ref)
    fwd, step = m.algorithmic_flops(1, 128, 128)
    assert fwd == pytest.approx(2 * 16384 * (27 * 64 + 576 * 64 + 64))          # 1.27 GFLOP, SURVEY 8a A9
    assert step == pytest.approx(3 * fwd - 2 * 16384 * 27 * 64)


def test_cnn3_at_the_configs1_shape_vs_oracle():
    """BASELINE configs[1] as written: the 3-layer CNN (width 64) on batch 64 x 128 x 128 x 3, one training step in the
    float32 arithmetic against oracle/cnn_ref.py on the same seeded inputs (a float32 CPU step of 3.7 GFLOP per patch:
    tens of seconds on the host).  Loss, logits, every gradient tensor, the clipped gradient norm."""
    st = cnn_ref.init_state(3, 1, 64, seed=21)
    g = torch.Generator().manual_seed(22)
    x = torch.randn(64, 128, 128, 3, generator=g)
    y = (torch.rand(64, 128, 128, generator=g) > 0.85).to(torch.uint8)
    y[:, :, 60:64] = 1
    xo, yo = unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)
    l32, lg32, g32, _ = cnn_ref.loss_and_grads(st, xo, yo)
    m = SimpleCNN(3, 1, 64).load_state_dict(st).train()
    loss = m.forward_backward(x, y)
    assert loss == pytest.approx(float(l32), abs=5e-6)
    want = lg32.permute(0, 2, 3, 1).reshape(-1).numpy()
    np.testing.assert_allclose(m.debug_tensor("logits"), want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))
    for k in g32:
        ref = g32[k].numpy()
        # two float32 summation orders over 64 x 16384 pixels: relative L2 at the 1e-5 level
        assert np.linalg.norm(m.grad(k) - ref) <= 5e-5 * np.linalg.norm(ref) + 1e-9, k
    total, _ = unet_ref.clip_coefficient(g32, 1.0)
    assert m.apply_gradients(lr=1e-4, weight_decay=1e-5) == pytest.approx(float(total), rel=1e-4)
