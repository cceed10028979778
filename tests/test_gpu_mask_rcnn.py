"""The detector assembly (``MaskRCNN``: SURVEY.md 8a row A11, BASELINE.json configs[3]) end to end on the GPU: a
builder-defined pipeline (no reference counterpart), so the checks are structural -- every stage is parity-tested on its
own (test_gpu_backbone / test_gpu_mask_head / test_gpu_detection_ops) -- plus the property that optimisation steps on a
fixed batch drive the summed loss down."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(rng, n=2, size=128):
    x = rng.standard_normal((n, size, size, 3)).astype(np.float32) * 0.1
    targets = []
    for i in range(n):
        boxes, masks = [], []
        for _ in range(2):
            w, h = rng.integers(20, 60, 2)
            x1, y1 = rng.integers(0, size - w), rng.integers(0, size - h)
            m = np.zeros((size, size), np.uint8)
            m[y1:y1 + h, x1:x1 + w] = 1
            x[i, y1:y1 + h, x1:x1 + w] += 2.0
            boxes.append([x1, y1, x1 + w, y1 + h]); masks.append(m)
        targets.append({"boxes": np.asarray(boxes, np.float32), "labels": np.ones(2, np.int64), "masks": np.stack(masks)})
    return x, targets


def test_anchor_grid_order():
    from rfi_toolbox_amd.models.mask_rcnn import _level_anchors
    a = _level_anchors(2, 3, 8, 16.0).reshape(2, 3, 4, 4)
    cx, cy = (a[..., 0] + a[..., 2]) / 2, (a[..., 1] + a[..., 3]) / 2
    assert np.allclose(cx[1, 2], 20.0) and np.allclose(cy[1, 2], 12.0)
    area = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    assert np.allclose(area[..., :3], 256.0, rtol=1e-5) and np.allclose(area[..., 3], 576.0)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_train_and_predict(dtype):
    import torch
    from rfi_toolbox_amd.models import MaskRCNN
    torch.manual_seed(0)
    det = MaskRCNN(2, 3, 16, 64, 128, seed=0).set_compute_dtype(dtype)
    x, targets = _batch(np.random.default_rng(1))
    first = det.train_step(x, targets, lr=2e-3, weight_decay=0.0, max_grad_norm=10.0)
    assert set(first) == {"loss_objectness", "loss_rpn_box_reg", "loss_classifier", "loss_box_reg", "loss_mask", "loss"}
    assert all(np.isfinite(v) for v in first.values())
    assert abs(first["loss_classifier"] - np.log(2)) < 0.3 and abs(first["loss_objectness"] - np.log(2)) < 0.3
    hist = [first["loss"]]
    for _ in range(14):
        hist.append(det.train_step(x, targets, lr=2e-3, weight_decay=0.0, max_grad_norm=10.0)["loss"])
    assert np.isfinite(hist).all()
    assert min(hist[-3:]) < 0.7 * hist[0], hist
    out = det.predict(x)
    assert len(out) == 2
    for o in out:
        k = len(o["boxes"])
        assert k <= det.max_det and o["boxes"].shape == (k, 4) and o["scores"].shape == (k,) and o["labels"].shape == (k,)
        assert o["masks"].shape == (k, 128, 128) and o["masks"].dtype == bool and o["rfi_mask"].shape == (128, 128)
        assert (o["boxes"][:, 0] >= 0).all() and (o["boxes"][:, 2] <= 128).all() and (np.diff(o["scores"]) <= 1e-6).all()
        assert (o["labels"] == 1).all()
        for b, m in zip(o["boxes"], o["masks"]):          # a pasted mask stays inside its (outward-rounded) box
            ys, xs = np.nonzero(m)
            if len(ys):
                assert xs.min() >= np.floor(b[0]) and xs.max() < np.ceil(b[2]) and ys.min() >= np.floor(b[1]) and ys.max() < np.ceil(b[3])


def test_paste_identity():
    from rfi_toolbox_amd.models.mask_rcnn import _paste
    prob = np.zeros((28, 28), np.float32)
    prob[:, 14:] = 1.0
    m = _paste(prob, [10, 20, 38, 48], 64, 64)
    assert m[20:48, 24:38].all() and not m[20:48, 10:24].any() and m.sum() == 28 * 14
