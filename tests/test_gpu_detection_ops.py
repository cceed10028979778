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
