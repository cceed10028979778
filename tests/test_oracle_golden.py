"""Pin the CPU oracle (oracle/) to the golden vectors captured from the reference.

CPU-only.  If these fail the oracle may not be used to judge the HIP path.
"""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import metrics_ref, preprocess_ref, unet_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _state(npz, tag):
    st = OrderedDict()
    pre = tag + "/"
    for k in npz.files:
        if k.startswith(pre):
            st[k[len(pre):]] = torch.from_numpy(npz[k].copy())
    return st


# ------------------------------------------------------------------ metrics
def test_metrics_match_reference(golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "metrics_expected.json")))
    inp = _load(golden_dir, "metrics_inputs.npz")
    assert len(exp) == 7
    for name, want in exp.items():
        got = metrics_ref.evaluate_segmentation(inp[f"{name}/pred"], inp[f"{name}/true"])
        for k in ("iou", "precision", "recall", "f1", "dice"):
            assert got[k] == pytest.approx(want[k], abs=1e-15), (name, k)


# ------------------------------------------------------------------ preprocessing
def test_patchify_known_answer(golden_dir):
    g = _load(golden_dir, "preprocess.npz")
    t = preprocess_ref.tile2d(np.arange(16).reshape(4, 4), 2)
    assert t.shape == (4, 2, 2) and tuple(g["e_shape"]) == (2, 2, 2, 2)
    np.testing.assert_array_equal(t[0], g["e_first"])          # [[0,1],[4,5]]
    np.testing.assert_array_equal(t[-1], g["e_last"])          # [[10,11],[14,15]]


def test_channel_extractors(golden_dir):
    g = _load(golden_dir, "preprocess.npz")
    for src, want in (("d_z", "d_z_ch"), ("d_zc", "d_zc_ch"), ("d_zz", "d_zz_ch")):
        got = preprocess_ref.channels_complex(g[src][None])[0]
        np.testing.assert_allclose(got, g[want], rtol=0, atol=1e-14)
    np.testing.assert_allclose(preprocess_ref.channels_real(np.abs(g["d_z"])[None])[0], g["d_r_ch"],
                               rtol=0, atol=1e-14)
    np.testing.assert_allclose(preprocess_ref.channels_real(np.abs(g["d_zc"])[None])[0], g["d_rc_ch"],
                               rtol=0, atol=1e-14)
    assert np.all(g["d_zc_ch"][..., 0] == 0)                   # constant patch -> zero gradient ch.


def test_create_dataset_complex(golden_dir):
    g = _load(golden_dir, "preprocess.npz")
    np.random.seed(7)
    img, lab = preprocess_ref.create_dataset(g["a_w"], g["a_m"], patch_size=64)
    assert img.shape == g["a_img"].shape == (4, 64, 64, 3) and img.dtype == np.float32
    np.testing.assert_allclose(img, g["a_img"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab, g["a_lab"])

    np.random.seed(8)
    img, lab = preprocess_ref.create_dataset(g["b_w"], g["b_m"], patch_size=32)
    assert img.shape == g["b_img"].shape
    np.testing.assert_allclose(img, g["b_img"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab, g["b_lab"])

    img, lab = preprocess_ref.create_dataset(g["b_w"], g["b_m"], patch_size=32,
                                             enable_augmentation=False, inference_mode=True)
    np.testing.assert_allclose(img, g["b2_img"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab, g["b2_lab"])
    assert lab.sum() == 0

    np.random.seed(9)
    img, lab = preprocess_ref.create_dataset(g["b_w"], g["b_m"], patch_size=32,
                                             augmentation_rotations=2, num_patches=10)
    assert len(img) == 10
    np.testing.assert_allclose(img, g["b3_img"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab, g["b3_lab"])


@pytest.mark.parametrize("tag,kw", [
    ("c_sqrt", dict(stretch_kind="SQRT")),
    ("c_log", dict(stretch_kind="LOG10", normalize_after_stretch=True)),
    ("c_none", dict(stretch_kind=None, normalize_before_stretch=False)),
])
def test_create_dataset_real(golden_dir, tag, kw):
    g = _load(golden_dir, "preprocess.npz")
    np.random.seed(10)
    img, lab = preprocess_ref.create_dataset(g["c_w"], None, patch_size=32, flag_sigma=5, **kw)
    assert img.shape == g[f"{tag}_img"].shape
    np.testing.assert_allclose(img, g[f"{tag}_img"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(lab, g[f"{tag}_lab"])


def test_bad_inputs_raise():
    with pytest.raises(ValueError):
        preprocess_ref.create_dataset(np.zeros((4, 4)))
    with pytest.raises(ValueError):
        preprocess_ref.create_dataset(np.ones((1, 1, 8, 8)), stretch_kind="CBRT", patch_size=8)


# ------------------------------------------------------------------ U-Net
def test_param_table_matches_reference_state_dict(golden_dir):
    g = _load(golden_dir, "unet_f4_b4_s32.npz")
    names = [str(n) for n in g["names"]]
    ent = unet_ref.unet_entries(3, 1, 4)
    assert [e[0] for e in ent] == names and len(names) == 136
    for name, shape, _ in ent:
        assert tuple(g[f"state0/{name}"].shape) == tuple(shape), name
    gb = _load(golden_dir, "unetbigger_f4_b2_s32.npz")
    entb = unet_ref.unet_entries(3, 1, 4, depth=5)
    assert [e[0] for e in entb] == [str(n) for n in gb["names"]]
    # parameter count of the flagship config
    n32 = sum(int(np.prod(s)) for _, s, k in unet_ref.unet_entries(3, 1, 32) if k == "param")
    assert n32 == 7_765_985


def test_forward_eval_and_train_f4(golden_dir):
    g = _load(golden_dir, "unet_f4_b4_s32.npz")
    st = _state(g, "state0")
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    with torch.no_grad():
        ev = unet_ref.forward(st, x, training=False)
        tr = unet_ref.forward(st, x, training=True, buffer_updates={})
    np.testing.assert_allclose(ev.numpy(), g["logits_eval0"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr.numpy(), g["logits_train1"], rtol=0, atol=5e-6)


def test_three_steps_f4_full_state(golden_dir):
    g = _load(golden_dir, "unet_f4_b4_s32.npz")
    lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
    st = _state(g, "state0")
    adam = unet_ref.new_adam_state(st)
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    y = torch.from_numpy(g["lab"]).float().unsqueeze(1)
    for s in (1, 2, 3):
        r = unet_ref.train_step(st, adam, x, y, lr=lr, betas=(b1, b2), eps=eps,
                                weight_decay=wd, clip=clip)
        assert r["loss"] == pytest.approx(float(g["losses"][s - 1]), abs=2e-6)
        assert r["grad_norm"] == pytest.approx(float(g["grad_norms"][s - 1]), rel=1e-4)
        if s == 1:
            for k, gr in r["grads"].items():          # golden grads are post-clip
                np.testing.assert_allclose((gr * r["clip_coef"]).numpy(), g[f"grad1/{k}"],
                                           rtol=0, atol=3e-6, err_msg=k)
        if s in (1, 3):
            for k, v in st.items():
                want = g[f"state{s}/{k}"]
                if k.endswith("num_batches_tracked"):
                    assert int(v) == int(want) == (2 * s if k.startswith("encoder") else s), k
                else:
                    # lr=1e-3: an Adam update is O(lr); sign flips of ~0 grads may move a
                    # conv bias that sits in front of a BatchNorm by up to ~lr per step
                    tol = 3e-3 if (k.endswith(".0.bias") or k.endswith(".3.bias")) else 1e-4 * s
                    np.testing.assert_allclose(v.numpy(), want, rtol=0, atol=tol, err_msg=k)
            with torch.no_grad():
                ev = unet_ref.forward(st, x, training=False)
            np.testing.assert_allclose(ev.numpy(), g[f"logits_eval{s}"], rtol=0, atol=2e-4)
    for k in adam["m"]:
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            continue
        np.testing.assert_allclose(adam["m"][k].numpy(), g[f"adam_m3/{k}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(adam["v"][k].numpy(), g[f"adam_v3/{k}"], rtol=0, atol=1e-7)


def test_f8_step1_and_trajectory(golden_dir):
    g = _load(golden_dir, "unet_f8_b4_s64.npz")
    st = _state(g, "state0")
    adam = unet_ref.new_adam_state(st)
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    y = torch.from_numpy(g["lab"]).float().unsqueeze(1)
    ious = {}
    for s in range(1, 41):
        r = unet_ref.train_step(st, adam, x, y, lr=1e-3, weight_decay=1e-5)
        if s == 1:
            np.testing.assert_allclose(r["logits"].numpy(), g["logits_train1"], rtol=0, atol=1e-5)
            for k in g.files:
                if k.startswith("grad1/"):
                    np.testing.assert_allclose((r["grads"][k[6:]] * r["clip_coef"]).numpy(), g[k],
                                               rtol=0, atol=3e-6, err_msg=k)
        assert r["loss"] == pytest.approx(float(g["losses"][s - 1]), abs=5e-3 if s > 3 else 1e-5), s
        if s % 10 == 0:
            with torch.no_grad():
                pred = unet_ref.predict_mask(unet_ref.forward(st, x, training=False))
            ious[s] = metrics_ref.evaluate_segmentation(pred, y)["iou"]
    for s, want in zip(g["iou_steps"], g["iou"]):
        assert abs(ious[int(s)] - float(want)) <= 1e-3, (s, ious[int(s)], want)
    # the reference-trained weights stored for the inference-parity test score the reference's IoU
    st40 = _state(g, "state40")
    with torch.no_grad():
        lg = unet_ref.forward(st40, x, training=False)
    np.testing.assert_allclose(lg.numpy(), g["logits_eval40"], rtol=0, atol=1e-5)
    assert metrics_ref.evaluate_segmentation(unet_ref.predict_mask(lg), y)["iou"] == pytest.approx(
        float(g["iou"][-1]), abs=1e-3)


def test_unet_bigger_variant(golden_dir):
    g = _load(golden_dir, "unetbigger_f4_b2_s32.npz")
    st = _state(g, "state0")
    assert unet_ref.infer_config(st) == (3, 1, 4, 5)
    adam = unet_ref.new_adam_state(st)
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    y = torch.from_numpy(g["lab"]).float().unsqueeze(1)
    r = unet_ref.train_step(st, adam, x, y, lr=1e-3, weight_decay=1e-5)
    np.testing.assert_allclose(r["logits"].numpy(), g["logits_train1"], rtol=0, atol=1e-5)
    assert r["loss"] == pytest.approx(float(g["losses"][0]), abs=2e-6)
    for k in g.files:
        if k.startswith("state1/"):
            np.testing.assert_allclose(st[k[7:]].numpy(), g[k], rtol=0, atol=3e-5, err_msg=k)


# ------------------------------------------------------------------ 3-layer CNN
def test_cnn3_oracle_matches_torch_module_fixture(golden_dir):
    """The builder-defined 3-layer CNN (no reference class exists): oracle/cnn_ref.py against the
    fixture produced by an nn.Sequential of the same layers driven by the reference's step."""
    from oracle import cnn_ref
    g = _load(golden_dir, "cnn3_c16_b4_s32.npz")
    st = _state(g, "state0")
    assert list(st.keys()) == [e[0] for e in cnn_ref.entries(3, 1, 16)] == [str(n) for n in g["names"]]
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    y = torch.from_numpy(g["lab"]).float().unsqueeze(1)
    np.testing.assert_allclose(cnn_ref.forward(st, x).numpy(), g["logits_eval0"], rtol=0, atol=1e-6)
    lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
    adam = cnn_ref.new_adam_state(st)
    for s in (1, 2, 3):
        r = cnn_ref.train_step(st, adam, x, y, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, clip=clip)
        assert r["loss"] == pytest.approx(float(g["losses"][s - 1]), abs=1e-6)
        assert r["grad_norm"] == pytest.approx(float(g["grad_norms"][s - 1]), rel=1e-5)
        if s == 1:
            coef = r["clip_coef"]
            for k, gr in r["grads"].items():
                np.testing.assert_allclose((gr * coef).numpy(), g[f"grad1/{k}"], rtol=0, atol=1e-6, err_msg=k)
        if s in (1, 3):
            for k, v in st.items():
                np.testing.assert_allclose(v.numpy(), g[f"state{s}/{k}"], rtol=0, atol=2e-6, err_msg=k)
    for k in st:
        np.testing.assert_allclose(adam["m"][k].numpy(), g[f"adam_m3/{k}"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(adam["v"][k].numpy(), g[f"adam_v3/{k}"], rtol=0, atol=1e-9)


# ------------------------------------------------------------------ U-Net variants
@pytest.mark.parametrize("name,kw", [("unet_overfit_f4_b2_s64.npz", {"head_sigmoid": True}),
                                     ("unet_leaky_f4_b2_s32.npz", {"negative_slope": 0.01})])
def test_unet_variants_oracle_matches_reference(golden_dir, name, kw):
    g = _load(golden_dir, name)
    st = _state(g, "state0")
    fn = unet_ref.variant_forward(**kw)
    x = unet_ref.nhwc_to_nchw(torch.from_numpy(g["img"]))
    y = torch.from_numpy(g["lab"]).float().unsqueeze(1)
    np.testing.assert_allclose(fn(st, x, training=False).numpy(), g["logits_eval0"], rtol=0, atol=2e-6)
    lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
    adam = unet_ref.new_adam_state(st)
    r = unet_ref.train_step(st, adam, x, y, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, clip=clip, forward_fn=fn)
    assert r["loss"] == pytest.approx(float(g["losses"][0]), abs=2e-6)
    np.testing.assert_allclose(r["logits"].numpy(), g["logits_train1"], rtol=0, atol=1e-5)
    for k, gr in r["grads"].items():
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            continue
        want = g[f"grad1/{k}"]
        got = (gr * r["clip_coef"]).numpy()
        assert np.linalg.norm(got - want) <= 1e-4 * np.linalg.norm(want) + 1e-9, k


def test_builder_defined_oracles_match_their_torch_modules():
    """The oracles of models the reference does not contain restate plain torch.nn modules:
    functional forward == module forward, state_dict order == module order."""
    import torch

    from oracle import mask_head_ref, resnet_unet_ref
    st = resnet_unet_ref.init_state(3, 1, 8, seed=1)
    mod = resnet_unet_ref.ResNetUNet(3, 1, 8)
    mod.load_state_dict(st)
    x = torch.randn(2, 3, 32, 32)
    assert float((mod.eval()(x) - resnet_unet_ref.forward(st, x)).abs().max()) == 0.0
    bu = {}
    out = resnet_unet_ref.forward(st, x, training=True, buffer_updates=bu)
    assert float((mod.train()(x) - out).abs().max()) <= 5e-6
    sd = mod.state_dict()
    assert max(float((sd[k].float() - bu[k].float()).abs().max()) for k in bu) <= 1e-6
    st = mask_head_ref.init_state(8, 1, 4, seed=2)
    mh = mask_head_ref.MaskHeadModule(8, 1, 4)
    mh.load_state_dict(st)
    x = torch.randn(3, 8, 14, 14)
    assert float((mh(x) - mask_head_ref.forward(st, x)).abs().max()) == 0.0
    y = (torch.rand(3, 1, 28, 28) > 0.5).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(mh(x), y)
    assert float(loss) == pytest.approx(float(mask_head_ref.mask_loss(mask_head_ref.forward(st, x), y)), abs=1e-7)
    st = mask_head_ref.rpn_init_state(8, 4, 1, seed=3)
    rp = mask_head_ref.RPNHeadModule(8, 4, 1)
    rp.load_state_dict(st)
    x = torch.randn(2, 8, 6, 5)
    cls, box = rp(x)
    out = mask_head_ref.rpn_forward(st, x)
    assert float((torch.cat([cls, box], 1).permute(0, 2, 3, 1) - out).abs().max()) == 0.0
    # the torch restatement of the RPN loss against the NumPy one (oracle/detection_ref.py) incl. its gradient
    import numpy as np

    from oracle import detection_ref
    rng = np.random.default_rng(0)
    labels = rng.choice(np.array([-1, -1, 0, 1], np.int8), 2 * 6 * 5 * 4)
    targets = rng.standard_normal((2 * 6 * 5 * 4, 4)).astype(np.float32) * 0.3
    o = out.detach().clone().requires_grad_(True)
    lo, lb = mask_head_ref.rpn_loss_torch(o, labels, targets, 4)
    (g,) = torch.autograd.grad(lo + lb, [o])
    wo, wb, wg = detection_ref.rpn_loss(out.detach().numpy().reshape(-1, 20), labels, targets, 4)
    assert float(lo) == pytest.approx(wo, rel=1e-5) and float(lb) == pytest.approx(wb, rel=1e-5)
    assert np.abs(g.numpy().reshape(-1, 20) - wg).max() <= 1e-6
