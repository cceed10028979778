"""Host-side bookkeeping of the detector (rfi_toolbox_amd/models/mask_rcnn.py) that needs no GPU."""
import numpy as np


def test_topk_selection_equals_the_stable_descending_sort():
    from rfi_toolbox_amd.models.mask_rcnn import _topk_desc_stable
    rng = np.random.default_rng(0)
    for m, k in ((4096, 200), (16, 200), (256, 200), (1024, 7), (1, 1)):
        sc = rng.standard_normal((5, m)).astype(np.float32)
        sc[:, ::7] = 0.5                       # ties
        sc[0, :] = 0
        if m > 8:
            sc[1, 3], sc[1, 4], sc[2, 5], sc[2, 6] = -0.0, 0.0, np.inf, -np.inf
        assert np.array_equal(_topk_desc_stable(sc, k), np.argsort(-sc, axis=1, kind="stable")[:, :k]), (m, k)


def test_anchor_grid_matches_the_oracle_grid():
    from oracle.mask_rcnn_ref import level_anchors
    from rfi_toolbox_amd.models.mask_rcnn import _level_anchors
    for h, w, s in ((2, 3, 8), (4, 4, 32), (1, 1, 64)):
        np.testing.assert_allclose(_level_anchors(h, w, s, 2.0 * s), level_anchors(h, w, s, 2.0 * s), rtol=1e-6, atol=1e-5)
