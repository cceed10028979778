"""The detector assembly (``MaskRCNN``: SURVEY.md 8a row A11, BASELINE.json configs[3]) end to end on the GPU: a
builder-defined pipeline (no reference counterpart).  Every stage is parity-tested on its own (test_gpu_backbone /
test_gpu_mask_head / test_gpu_detection_ops); here the ASSEMBLED training step is checked against the assembled oracle
(oracle/mask_rcnn_ref.py) -- five losses, the gradient norms of the four parameter sets, and every discrete decision
(sampled anchors, proposals, sampled RoIs, mask targets) -- at reduced and at full width (ResNet-50, 256 pyramid channels,
1024-wide box head), plus the property that optimisation steps on a fixed batch drive the summed loss down."""
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


def _states(det):
    return {"backbone": det.backbone.state_dict(), "rpn": det.rpn.state_dict(), "box": det.box.state_dict(), "mask": det.mask.state_dict()}


@pytest.mark.parametrize("widths", [(16, 64, 128), (64, 256, 1024)], ids=["reduced", "resnet50_fpn256"])
def test_assembled_step_against_the_oracle(widths):
    import torch
    from oracle.mask_rcnn_ref import MaskRCNNRef
    from rfi_toolbox_amd.models import MaskRCNN
    torch.manual_seed(3)
    det = MaskRCNN(2, 3, *widths, seed=7)
    det.keep_trace = True
    ref = MaskRCNNRef(2, 3, *widths).load(_states(det))
    x, targets = _batch(np.random.default_rng(1))
    got = det.train_step(x, targets, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)        # (lr 0: the weights stay what the oracle holds)
    tr = det.last_trace
    # continuous quantities, on the decisions the device step took
    want, wtr = ref.step(x, targets, decisions=tr)
    for k in want:
        assert abs(got[k] - want[k]) <= 5e-4 * max(1.0, abs(want[k])), (k, got[k], want[k])
    for k, v in wtr["grad_norms"].items():
        assert abs(tr["grad_norms"][k] - v) <= 3e-3 * v, (k, tr["grad_norms"][k], v)
    # discrete decisions: a free oracle run from the same seed takes the same ones
    free, ftr = ref.step(x, targets, sampler=(det.seed, det.sample_step - 1), grads=False)
    assert np.array_equal(ftr["rpn_labels"], tr["rpn_labels"])
    assert tr["num_sampled"] == int((ftr["rpn_labels"] >= 0).sum())
    np.testing.assert_allclose(ftr["rpn_targets"], tr["rpn_targets"], rtol=1e-5, atol=1e-6)
    assert [len(p) for p in ftr["proposals"]] == [len(p) for p in tr["proposals"]]
    for a, b in zip(ftr["proposals"], tr["proposals"]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3)
    np.testing.assert_allclose(ftr["rois"], tr["rois"], rtol=0, atol=2e-3)               # image-major, positives of an image first
    assert np.array_equal(ftr["roi_labels"], tr["roi_labels"]) and np.array_equal(ftr["roi_gt"], tr["roi_gt"])
    assert np.array_equal(ftr["roi_levels"], tr["roi_levels"])
    np.testing.assert_allclose(ftr["roi_targets"], tr["roi_targets"], rtol=1e-4, atol=1e-5)
    for k in free:
        assert abs(got[k] - free[k]) <= 5e-4 * max(1.0, abs(free[k])), (k, got[k], free[k])


def test_batched_matcher_and_nms_match_the_per_image_oracle():
    from oracle import detection_ref
    from rfi_toolbox_amd.models import detection_ops as ops
    rng = np.random.default_rng(0)
    anchors = np.concatenate([rng.uniform(0, 90, (300, 2)), rng.uniform(0, 90, (300, 2))], 1).astype(np.float32)
    anchors[:, 2:] = anchors[:, :2] + rng.uniform(4, 40, (300, 2)).astype(np.float32)
    gts = []
    for g in (3, 0, 1, 5):
        b = rng.uniform(0, 80, (g, 2)).astype(np.float32)
        gts.append(np.concatenate([b, b + rng.uniform(8, 40, (g, 2)).astype(np.float32)], 1))
    lab, mi, tg = ops.anchor_match_batched(anchors, gts)
    for i, g in enumerate(gts):
        l0, m0, t0 = detection_ref.anchor_match(anchors, g)
        assert np.array_equal(lab[i], l0) and np.array_equal(mi[i], m0)
        np.testing.assert_allclose(tg[i], t0, rtol=1e-5, atol=1e-6)
    # per-image boxes with counts (the RoI stage), thresholds 0.5 / 0.5 without the low-quality rule
    per = np.stack([anchors[rng.permutation(300)] for _ in gts])
    counts = [300, 17, 0, 256]
    lab, mi, tg = ops.anchor_match_batched(per, gts, 0.5, 0.5, False, anchor_counts=counts)
    for i, g in enumerate(gts):
        l0, m0, t0 = detection_ref.anchor_match(per[i, :counts[i]], g, 0.5, 0.5, False)
        assert np.array_equal(lab[i, :counts[i]], l0) and np.array_equal(mi[i, :counts[i]], m0) and (lab[i, counts[i]:] == -2).all()
        np.testing.assert_allclose(tg[i, :counts[i]], t0, rtol=1e-5, atol=1e-6)
    # batched NMS: sets in descending score order
    K = 200
    sets = np.zeros((5, K, 4), np.float32)
    cnt = np.array([200, 1, 0, 77, 130], np.int32)
    for s in range(5):
        p = rng.uniform(0, 100, (K, 2)).astype(np.float32)
        sets[s] = np.concatenate([p, p + rng.uniform(5, 50, (K, 2)).astype(np.float32)], 1)
    keep = ops.nms_batched(sets, cnt, 0.7)
    for s in range(5):
        want = detection_ref.nms(sets[s, :cnt[s]], -np.arange(cnt[s], dtype=np.float32), 0.7) if cnt[s] else np.zeros(0, np.int64)
        assert np.array_equal(np.flatnonzero(keep[s]), np.sort(want)), s


def test_mask_targets_kernel_against_the_roi_align_oracle():
    import ctypes as C
    from oracle import detection_ref
    from rfi_toolbox_amd._lib import check, lib
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    rng = np.random.default_rng(2)
    masks = (rng.random((3, 40, 48)) > 0.6).astype(np.uint8)
    masks[1] = 0
    masks[1, 10:30, 5:25] = 1
    rois = np.array([[0, 2.5, 3.0, 30.0, 33.0], [1, 5.0, 10.0, 25.0, 30.0], [2, -4.0, -2.0, 20.0, 50.0], [1, 0.0, 0.0, 48.0, 40.0]], np.float32)
    dm, dr = ctx.to_device(masks), ctx.to_device(rois)
    out = ctx.empty((len(rois), 28, 28), np.uint8)
    check(lib.rfi_op_mask_targets(ctx.handle, C.c_void_p(dm.ptr), 3, 40, 48, C.c_void_p(dr.ptr), len(rois), 28, 28, 2, C.c_void_p(out.ptr)))
    ctx.synchronize()
    got = out.numpy()
    for j, r in enumerate(rois):
        roi = np.concatenate([[0.0], r[1:]]).astype(np.float32)[None]
        v = detection_ref.roi_align(masks[int(r[0])].astype(np.float32)[None, :, :, None], roi, 1.0, (28, 28), 2, False)[0, :, :, 0]
        clear = np.abs(v - 0.5) > 1e-4                                   # (a value within rounding of the threshold may fall either way)
        assert np.array_equal(got[j][clear], (v >= 0.5)[clear].astype(np.uint8)), j
    assert got[1][:26, :26].all()                                         # the RoI is the rectangle (its far edge interpolates to the outside)


def test_assembled_step_is_bitwise_reproducible():
    """No float atomics anywhere in the step (the RoIAlign gradient is a gather, every reduction has a fixed order): the
    same weights, batch and sampler seed give bit-identical losses and gradient norms, with the weight-gradient kernels of
    four models sharing the side stream, the slab workspace and the event rings."""
    import torch
    from rfi_toolbox_amd.models import MaskRCNN
    torch.manual_seed(3)
    det = MaskRCNN(2, 3, 16, 64, 128, seed=7)
    det.keep_trace = True
    x, targets = _batch(np.random.default_rng(1), n=4)
    watch = [e[0] for e in det.backbone._entries if e[0].endswith("weight") and ("conv" in e[0] or "downsample.0" in e[0] or "blocks" in e[0])][::5]
    runs = []
    for _ in range(4):
        det.sample_step = 11
        losses = det.train_step(x, targets, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
        # (the backbone's weight gradients run on the side stream next to its main chain: whole tensors, not only their norm)
        grads = {k: det.backbone.grad(k).copy() for k in watch}
        runs.append((losses, dict(det.last_trace["grad_norms"]), det.last_trace["rois"].copy(), grads))
    for losses, norms, rois, grads in runs[1:]:
        assert losses == runs[0][0] and norms == runs[0][1] and np.array_equal(rois, runs[0][2])
        assert all(np.array_equal(grads[k], runs[0][3][k]) for k in watch)


def test_data_parallel_step_on_a_rank_without_foreground():
    """grad_sync = 2 (emulated exchange, rfi_comm_emulate): a rank whose images hold no box has no foreground RoI, hence no
    mask-branch gradient -- it must still take part in the all-reduce of the mask head's gradients (with zeros) and apply the
    averaged update like every other rank; skipping both would hang the collective of a real job.  With targets present the
    emulated step must equal the plain one bit for bit (every element exchanged once, scaled back by 1 / 2)."""
    import torch
    from rfi_toolbox_amd.models import MaskRCNN
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    x, targets = _batch(np.random.default_rng(5))
    empty = [{"boxes": np.zeros((0, 4), np.float32), "labels": np.zeros(0, np.int64), "masks": np.zeros((0, 128, 128), np.uint8)} for _ in targets]
    try:
        res = []
        for world in (0, 2):
            ctx.comm_emulate(world)
            torch.manual_seed(4)
            det = MaskRCNN(2, 3, 16, 64, 128, seed=9)
            det.grad_sync = max(world, 1)
            det.sample_step = 3
            losses = det.train_step(x, targets, lr=1e-3)
            res.append((losses, dict(det.last_trace["grad_norms"]), {k: v.clone() for k, v in det.mask.state_dict().items()}))
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], res[1][2][k]), k
        # the rank without a box: the mask head is stepped (zero gradients: only the L2 term moves it)
        ctx.comm_emulate(2)
        before = {k: v.clone() for k, v in det.mask.state_dict().items()}
        losses = det.train_step(x, empty, lr=1e-3, weight_decay=1e-2)
        assert losses["loss_mask"] == 0.0 and np.isfinite(losses["loss"])
        assert det.last_trace["grad_norms"]["mask"] == 0.0
        after = det.mask.state_dict()
        assert any(not torch.equal(before[k], after[k]) for k in before)          # weight decay acted: the step was applied
        assert all(torch.isfinite(v).all() for v in after.values())
    finally:
        ctx.comm_emulate(0)
