"""-m gpu: RoIAlign and the FPN top-down merge (BASELINE configs[3]) against the NumPy oracle of
the published algorithms (oracle/detection_ref.py; parity unpinned by the reference, which has no detector).  float32
kernels vs float64 loops: relative max error <= 1e-5 forward, 2e-5 for the atomically accumulated backward."""
import numpy as np
import pytest

from oracle import detection_ref as ref
from rfi_toolbox_amd.models import detection_ops as ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _rois(rng, n_img, size, count):
    x1 = rng.uniform(-4, size * 0.8, count)
    y1 = rng.uniform(-4, size * 0.8, count)
    w = rng.uniform(0.3, size * 0.7, count)
    h = rng.uniform(0.3, size * 0.7, count)
    r = np.stack([rng.integers(0, n_img, count).astype(np.float64), x1, y1, x1 + w, y1 + h], 1)
    r[0] = [0, 0, 0, size, size]                         # the whole image
    r[1] = [n_img - 1, size - 2.0, size - 2.0, size + 6.0, size + 9.0]     # hangs over the border
    r[2, 1:] = [3.2, 5.7, 3.25, 5.75]                    # tiny: clamped to one pixel when not aligned
    return r


@pytest.mark.parametrize("out,sr,aligned,scale", [((7, 7), 2, False, 0.25), ((14, 14), 2, False, 0.125), ((7, 7), 0, True, 0.5),
                                                  ((3, 5), 3, True, 1.0)])
def test_roi_align_forward_backward(out, sr, aligned, scale):
    rng = np.random.default_rng(5)
    n, size, c = 2, 32, 8
    x = rng.standard_normal((n, size, size, c)).astype(np.float32)
    rois = _rois(rng, n, size / scale, 12).astype(np.float32)
    got = ops.roi_align(x, rois, scale, out, sr, aligned)
    want = ref.roi_align(x, rois, scale, out, sr, aligned)
    assert got.shape == want.shape and _rel(got, want) <= 1e-5
    dout = rng.standard_normal(want.shape).astype(np.float32)
    gdx = ops.roi_align_backward(dout, x.shape, rois, scale, sr, aligned)
    wdx = ref.roi_align_backward(dout, x.shape, rois, scale, out, sr, aligned)
    assert _rel(gdx, wdx) <= 2e-5
    # adjointness: <roi_align(x), dout> == <x, roi_align_backward(dout)>
    assert np.vdot(got.astype(np.float64), dout) == pytest.approx(np.vdot(x.astype(np.float64), gdx), rel=1e-4)


def test_roi_align_empty_and_bad_batch_index():
    x = np.ones((1, 8, 8, 4), np.float32)
    assert ops.roi_align(x, np.zeros((0, 5), np.float32), 1.0, (2, 2)).shape == (0, 2, 2, 4)
    got = ops.roi_align(x, np.array([[3, 0, 0, 4, 4], [0, 0, 0, 4, 4]], np.float32), 1.0, (2, 2))
    assert np.all(got[0] == 0) and np.allclose(got[1], 1.0)      # an out-of-range image index yields zeros, not a fault


@pytest.mark.parametrize("shape", [(2, 16, 16, 8), (1, 7, 9, 4), (3, 2, 2, 12)])
def test_fpn_merge_forward_backward(shape):
    rng = np.random.default_rng(6)
    n, h, w, c = shape
    lat = rng.standard_normal(shape).astype(np.float32)
    top = rng.standard_normal((n, (h + 1) // 2, (w + 1) // 2, c)).astype(np.float32)
    np.testing.assert_array_equal(ops.fpn_merge(lat, top), ref.fpn_merge(lat, top).astype(np.float32))
    dout = rng.standard_normal(shape).astype(np.float32)
    dlat, dtop = ops.fpn_merge_backward(dout)
    np.testing.assert_array_equal(dlat, dout)
    assert _rel(dtop, ref.fpn_merge_backward_top(dout)) <= 1e-6
    with pytest.raises(ValueError):
        ops.fpn_merge(lat, np.zeros((n, (h + 1) // 2 + 1, (w + 1) // 2, c), np.float32))


def _anchors(rng, n):
    x1, y1 = rng.uniform(0, 200, n), rng.uniform(0, 200, n)
    return np.stack([x1, y1, x1 + rng.uniform(4, 120, n), y1 + rng.uniform(4, 120, n)], 1).astype(np.float32)


def test_box_decode():
    rng = np.random.default_rng(1)
    anchors = _anchors(rng, 36)
    deltas = (rng.standard_normal((36 * 50, 4)) * np.array([0.5, 0.5, 1.5, 1.5])).astype(np.float32)
    deltas[3, 2:] = 9.0                                # beyond the clamp
    got = ops.decode_boxes(anchors, deltas)
    want = ref.decode_boxes(anchors, deltas)
    assert _rel(got, want) <= 2e-6
    got = ops.decode_boxes(anchors, deltas, image_size=(256, 300))
    want = ref.decode_boxes(anchors, deltas, image_size=(256, 300))
    assert np.abs(got - want).max() <= 1e-3 and got.min() >= 0 and got[:, 0::2].max() <= 300 and got[:, 1::2].max() <= 256


@pytest.mark.parametrize("n,thr", [(1, 0.5), (63, 0.3), (64, 0.7), (65, 0.5), (700, 0.7), (2000, 0.7)])
def test_nms(n, thr):
    rng = np.random.default_rng(n)
    centres = rng.uniform(20, 230, (max(n // 6, 1), 2))                  # clusters of overlapping boxes
    c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 6, (n, 2))
    wh = rng.uniform(15, 60, (n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rng.random(n).astype(np.float32)
    scores[: n // 10] = scores[0]                                       # ties: stable order decides
    got = ops.nms(boxes, scores, thr)
    want = ref.nms(boxes, scores, thr)
    # an IoU within float32 rounding of the threshold may fall on either side; none of these seeds has one
    np.testing.assert_array_equal(got, want)
    assert ops.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).size == 0


@pytest.mark.parametrize("p,a", [(37, 4), (64 * 64, 12), (2 * 50 * 76, 4)])
def test_rpn_loss_forward_backward(p, a):
    rng = np.random.default_rng(p)
    head = (rng.standard_normal((p, 5 * a)) * 1.5).astype(np.float32)
    labels = rng.choice(np.array([-1, -1, -1, 0, 0, 1], np.int8), p * a)
    targets = (rng.standard_normal((p * a, 4)) * 0.4).astype(np.float32)
    targets[::7] = head[:, a:].reshape(-1, 4)[::7] + 0.05              # inside the quadratic zone of smooth L1
    lo, lb, g = ops.rpn_loss(head, labels, targets, a)
    wo, wb, wg = ref.rpn_loss(head, labels, targets, a)
    assert lo == pytest.approx(wo, rel=2e-6) and lb == pytest.approx(wb, rel=2e-6)
    assert _rel(g, wg) <= 5e-6
    # nothing sampled: zero loss, zero gradient
    lo, lb, g = ops.rpn_loss(head, np.full(p * a, -1, np.int8), targets, a)
    assert lo == 0.0 and lb == 0.0 and not g.any()


@pytest.mark.parametrize("n,g", [(500, 0), (3000, 1), (20000, 7), (257, 40)])
def test_anchor_match_and_encode(n, g):
    rng = np.random.default_rng(n + g)
    anchors = _anchors(rng, n)
    gt = _anchors(rng, g) if g else np.zeros((0, 4), np.float32)
    if g:
        anchors[5] = gt[0]                                    # an exact hit (IoU 1)
    lab, matched, targets = ops.anchor_match(anchors, gt)
    wl, wm, wt = ref.anchor_match(anchors, gt)
    # the device contracts a*b+c into FMAs where NumPy rounds twice: an IoU within 1 ulp of a threshold may differ
    diff = np.flatnonzero(lab != wl)
    assert len(diff) <= max(1, n // 5000), (len(diff), n)
    same = lab == wl
    np.testing.assert_array_equal(matched[same & (lab == 1)], wm[same & (lab == 1)])
    assert (matched[lab != 1] == -1).all()
    pos = same & (lab == 1)
    if pos.any():
        assert np.abs(targets[pos] - wt[pos]).max() <= 2e-5 * max(1.0, np.abs(wt[pos]).max())
    assert not targets[lab != 1].any()
    if g:
        assert lab[5] == 1 and matched[5] == 0
        for k in range(g):                                   # every ground truth keeps at least one positive anchor
            assert (matched == k).any() or (wm == k).sum() == 0


def test_roi_align_backward_gather_form_matches_the_oracle_and_is_reproducible():
    """rfi_op_roi_align_backward_sorted: RoIs sorted by image, every element of dx written once, no atomics."""
    import ctypes as C
    from oracle import detection_ref
    from rfi_toolbox_amd._lib import check, lib
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    rng = np.random.default_rng(5)
    for (n, h, w, c, res, scale, R) in ((3, 16, 16, 8, 7, 0.25, 40), (2, 8, 12, 4, 14, 0.125, 25), (2, 32, 32, 16, 7, 0.25, 64), (1, 4, 4, 4, 7, 1.0 / 32, 9)):
        img = np.sort(rng.integers(0, n, R)).astype(np.float32)
        size = h / scale
        x1, y1 = rng.uniform(-0.1 * size, 0.8 * size, R), rng.uniform(-0.1 * size, 0.8 * size, R)
        bw, bh = rng.uniform(0.02 * size, 0.6 * size, R), rng.uniform(0.02 * size, 0.6 * size, R)
        rois = np.stack([img, x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        rois[0, 1:] = [0.0, 0.0, size, size]                                    # the whole map
        rois[-1, 1:] = [size * 0.9, size * 0.9, size * 1.3, size * 1.2]         # sticks out of the map
        dout = rng.standard_normal((R, res, res, c)).astype(np.float32)
        want = detection_ref.roi_align_backward(dout, (n, h, w, c), rois, scale, (res, res), 2, False)
        dd, dr = ctx.to_device(dout), ctx.to_device(rois)
        outs = []
        for _ in range(2):
            dx = ctx.to_device(np.full((n, h, w, c), 7.0, np.float32))           # (stale contents must be overwritten)
            check(lib.rfi_op_roi_align_backward_sorted(ctx.handle, C.c_void_p(dd.ptr), n, h, w, c, C.c_void_p(dr.ptr), R, float(scale), res, res,
                                                       2, 0, C.c_void_p(dx.ptr)))
            ctx.synchronize()
            outs.append(dx.numpy())
        assert np.array_equal(outs[0], outs[1])
        np.testing.assert_allclose(outs[0], want, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(want).max()))
