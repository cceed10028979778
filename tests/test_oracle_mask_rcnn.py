"""The assembled Mask R-CNN oracle (oracle/mask_rcnn_ref.py) checked against the per-stage oracles it
is assembled from, on the CPU: the differentiable RoIAlign against detection_ref.roi_align, the anchor grid, determinism of
the step under a seed, and replay of recorded decisions."""
import numpy as np
import torch

from oracle import detection_ref, mask_rcnn_ref


def _batch(rng, n=2, size=64):
    x = rng.standard_normal((n, size, size, 3)).astype(np.float32) * 0.1
    targets = []
    for i in range(n):
        boxes, masks = [], []
        for _ in range(2):
            w, h = rng.integers(12, 30, 2)
            x1, y1 = rng.integers(0, size - w), rng.integers(0, size - h)
            m = np.zeros((size, size), np.uint8)
            m[y1:y1 + h, x1:x1 + w] = 1
            x[i, y1:y1 + h, x1:x1 + w] += 2.0
            boxes.append([x1, y1, x1 + w, y1 + h]); masks.append(m)
        targets.append({"boxes": np.asarray(boxes, np.float32), "labels": np.ones(2, np.int64), "masks": np.stack(masks)})
    return x, targets


def test_roi_align_torch_matches_the_numpy_oracle():
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((2, 9, 11, 5)).astype(np.float32)                 # NHWC
    rois = np.array([[0, 1.0, 2.0, 30.0, 20.0], [1, -3.0, -2.0, 8.0, 9.0], [1, 10.0, 4.0, 10.5, 4.2], [0, 20.0, 15.0, 60.0, 50.0]], np.float32)
    for res, scale in ((7, 0.25), (14, 0.25), (3, 1.0 / 8)):
        want = detection_ref.roi_align(feat, rois, scale, (res, res), 2, False)                       # (R, res, res, C)
        got = mask_rcnn_ref.roi_align_torch(torch.as_tensor(feat).permute(0, 3, 1, 2), rois, scale, res).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_anchor_grid():
    a = mask_rcnn_ref.level_anchors(2, 3, 8, 16.0).reshape(2, 3, 4, 4)
    cx, cy = (a[..., 0] + a[..., 2]) / 2, (a[..., 1] + a[..., 3]) / 2
    assert np.allclose(cx[1, 2], 20.0) and np.allclose(cy[1, 2], 12.0)
    area = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    assert np.allclose(area[..., :3], 256.0, rtol=1e-5) and np.allclose(area[..., 3], 576.0)


def test_step_is_a_function_of_weights_batch_and_seed():
    torch.manual_seed(0)
    ref = mask_rcnn_ref.MaskRCNNRef(2, 3, 8, 16, 32)
    x, targets = _batch(np.random.default_rng(1))
    l1, t1 = ref.step(x, targets, sampler=(5, 0))
    l2, t2 = ref.step(x, targets, sampler=(5, 0))
    assert l1 == l2 and all(np.isfinite(v) for v in l1.values())
    assert abs(l1["loss_objectness"] - np.log(2)) < 0.3 and abs(l1["loss_classifier"] - np.log(2)) < 0.3
    assert (t1["rpn_labels"] == t2["rpn_labels"]).all() and np.array_equal(t1["rois"], t2["rois"])
    assert (t1["rpn_labels"] >= 0).sum(1).max() <= 256 and (t1["rpn_labels"] == 1).sum(1).max() <= 128
    assert len(t1["rois"]) <= 2 * 128 and set(np.unique(t1["roi_labels"])) <= {0, 1}
    assert set(t1["grad_norms"]) == {"backbone", "rpn", "box", "mask"} and all(v > 0 for v in t1["grad_norms"].values())
    _, t4 = ref.step(x, targets, sampler=(5, 1), grads=False)             # another step counter: another sample
    assert not np.array_equal(t1["rpn_labels"], t4["rpn_labels"])
    # replaying the recorded decisions (no generator) reproduces the losses
    l3, _ = ref.step(x, targets, decisions=t1, grads=False)
    for k in l1:
        assert abs(l1[k] - l3[k]) <= 1e-6 * max(1.0, abs(l1[k])), k
    # the mask targets of a rectangular instance are the rectangle seen from the RoI
    fg = np.flatnonzero(t1["roi_labels"] > 0)
    if len(fg):
        assert t1["mask_targets"].shape == (len(fg), 28, 28) and t1["mask_targets"].max() == 1


def test_counter_based_sampler_and_level_rule():
    """The sampler's preference order is ascending (Philox word, index); the level rule in comparison form equals
    clip(floor(4 + log2(sqrt(area) / (size / 2))), 2, 5) - 2 away from the level boundaries."""
    from oracle.synth_ref import philox4x32_10
    idx = np.array([7, 3, 100, 42, 5], np.int64)
    got = mask_rcnn_ref.sample_order(idx, 2, 1, (9 << 32) | 5, 11)
    r = [int(philox4x32_10([i], [2], [1], [11], 5, 9)[0][0]) for i in idx]
    assert list(got) == [i for _, i in sorted(zip(r, idx))]
    assert len(mask_rcnn_ref.sample_order([], 0, 0, 1, 0)) == 0
    rng = np.random.default_rng(0)
    b = rng.uniform(0, 100, (500, 2)).astype(np.float32)
    boxes = np.concatenate([b, b + rng.uniform(1, 120, (500, 2)).astype(np.float32)], 1)
    area = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), 1e-6).astype(np.float64)
    k = np.floor(4 + np.log2(np.sqrt(area) / 64.0))
    want = np.clip(k, 2, 5).astype(int) - 2
    frac = np.abs(np.log2(np.sqrt(area) / 64.0) - np.round(np.log2(np.sqrt(area) / 64.0)))
    clear = frac > 1e-4
    assert np.array_equal(mask_rcnn_ref.roi_levels(boxes, 128)[clear], want[clear]) and clear.sum() > 450
