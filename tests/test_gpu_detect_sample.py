"""-m gpu: the detector's box bookkeeping on the device (csrc/detect_sample.hip, the multi-level RoIAlign of
csrc/detect_kernels.hip) against NumPy statements of the same rules -- sort, counter-based samplers,
per-level top-k + decode, post-NMS selection, RoI lists + pyramid levels, multi-level RoIAlign both ways.  Builder-defined
(the reference has no detector): the oracle is oracle/mask_rcnn_ref.py / oracle/detection_ref.py, parity unpinned."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx():
    from rfi_toolbox_amd.runtime import Context
    return Context.get(0)


def _P(d):
    return C.c_void_p(d.ptr)


def test_segmented_sort_of_u64_keys():
    from rfi_toolbox_amd._lib import check, lib
    ctx = _ctx()
    rng = np.random.default_rng(0)
    for segs, stride in ((3, 2), (5, 64), (7, 1024), (4, 8192)):
        k = rng.integers(0, 2 ** 63, (segs, stride), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (segs, stride), dtype=np.uint64)
        k[0, : stride // 2] = np.uint64(0xFFFFFFFFFFFFFFFF)                      # padding keys and ties
        k[-1, :] = k[-1, 0]
        d = ctx.to_device(k)
        check(lib.rfi_op_segsort_u64(ctx.handle, _P(d), segs, stride))
        assert np.array_equal(d.numpy(), np.sort(k, axis=1)), (segs, stride)


def test_rpn_sampler_against_the_counter_based_oracle():
    from oracle.mask_rcnn_ref import sample_order
    from rfi_toolbox_amd._lib import check, lib
    ctx = _ctx()
    rng = np.random.default_rng(1)
    B, off = 4, np.array([0, 700, 1000, 1100], np.int32)
    n = int(off[-1])
    labels = rng.choice([-1, 0, 1], size=(B, n), p=[0.1, 0.8, 0.1]).astype(np.int8)
    labels[1] = np.where(labels[1] == 1, 0, labels[1])                          # an image without positives
    labels[2, 40:] = -1                                                         # ... and one with fewer candidates than the batch
    labels[3] = np.where(rng.random(n) < 0.5, 1, 0)                              # more positives than the cap
    targets = rng.standard_normal((B, n, 4)).astype(np.float32)
    seed, step, batch, max_pos = (77 << 32) | 5, 9, 64, 32
    stride = 2048
    dl, dt = ctx.to_device(labels), ctx.to_device(targets)
    keys, cnt = ctx.empty((B, stride), np.uint64), ctx.to_device(np.zeros(4, np.int32))
    lev_l = [ctx.empty((B * int(off[l + 1] - off[l]),), np.int8) for l in range(3)]
    lev_t = [ctx.empty((B * int(off[l + 1] - off[l]), 4), np.float32) for l in range(3)]
    check(lib.rfi_op_sample_keys(ctx.handle, _P(dl), B, n, None, seed, step, 0, _P(keys), stride))
    check(lib.rfi_op_segsort_u64(ctx.handle, _P(keys), B, stride))
    pl, pt = (C.c_void_p * 3)(*[a.ptr for a in lev_l]), (C.c_void_p * 3)(*[a.ptr for a in lev_t])
    check(lib.rfi_op_rpn_sample_apply(ctx.handle, _P(keys), B, n, stride, batch, max_pos, _P(dl), _P(dt), 3, off.ctypes.data_as(C.c_void_p),
                                      pl, pt, _P(cnt)))
    got = np.concatenate([a.numpy().reshape(B, -1) for a in lev_l], 1)
    got_t = np.concatenate([a.numpy().reshape(B, -1, 4) for a in lev_t], 1)
    want = labels.copy()
    for i in range(B):
        pos, neg = np.flatnonzero(labels[i] == 1), np.flatnonzero(labels[i] == 0)
        npos = min(len(pos), max_pos)
        want[i, sample_order(pos, i, 0, seed, step)[npos:]] = -1
        want[i, sample_order(neg, i, 1, seed, step)[batch - npos:]] = -1
    assert np.array_equal(got, want)
    assert np.array_equal(got_t, targets)
    assert int(cnt.numpy()[0]) == int((want >= 0).sum())
    assert (want[3] == 1).sum() == max_pos and (want[2] >= 0).sum() == (labels[2] >= 0).sum() < batch


def test_topk_decode_and_post_nms_selection():
    from oracle import detection_ref
    from rfi_toolbox_amd._lib import check, lib
    from rfi_toolbox_amd.models.mask_rcnn import _level_anchors, _topk_desc_stable
    ctx = _ctx()
    rng = np.random.default_rng(2)
    B, K, L = 3, 200, 2
    dims = [(16, 16, 8), (4, 4, 32)]                                             # (h, w, stride): 1024 and 64 anchors
    boxes = ctx.to_device(np.full((B, L, K, 4), np.nan, np.float32))
    scores, counts = ctx.to_device(np.full((B, L, K), np.nan, np.float32)), ctx.empty((B, L), np.int32)
    want_b, want_s, want_c = np.zeros((B, L, K, 4), np.float32), np.full((B, L, K), -np.inf, np.float32), np.zeros((B, L), np.int32)
    for lvl, (hl, wl, st) in enumerate(dims):
        P_, A = hl * wl, 4
        head = rng.standard_normal((B, P_, 5 * A)).astype(np.float32)
        head[:, :, 4:] *= 0.5
        head[0, ::3, :4] = 0.25                                                    # ties
        head[1, 5, 4 + 2] = -30.0                                                  # a box that collapses (width under min_size)
        anchors = _level_anchors(hl, wl, st, 2.0 * st)
        stride = max(2, 1 << int(P_ * A - 1).bit_length())
        dh, da, keys = ctx.to_device(head), ctx.to_device(anchors), ctx.empty((B, stride), np.uint64)
        check(lib.rfi_op_topk_keys(ctx.handle, _P(dh), B, P_, A, _P(keys), stride))
        check(lib.rfi_op_segsort_u64(ctx.handle, _P(keys), B, stride))
        check(lib.rfi_op_topk_decode(ctx.handle, _P(keys), B, stride, P_, A, K, _P(dh), _P(da), 128.0, 128.0, 1e-2, _P(boxes), _P(scores),
                                     _P(counts), L, lvl))
        sc, dl = head[:, :, :A].reshape(B, -1), head[:, :, A:].reshape(B, -1, 4)
        top = _topk_desc_stable(sc, K)
        k = top.shape[1]
        for i in range(B):
            bx = detection_ref.decode_boxes(anchors[top[i]], dl[i, top[i]], image_size=(128, 128))
            ok = ((bx[:, 2] - bx[:, 0]) >= 1e-2) & ((bx[:, 3] - bx[:, 1]) >= 1e-2)
            first = np.argsort(~ok, kind="stable")
            want_b[i, lvl, :k] = bx[first]
            want_s[i, lvl, :k] = np.where(ok, sc[i, top[i]], -np.inf)[first]
            want_c[i, lvl] = ok.sum()
    gb, gs, gc = boxes.numpy(), scores.numpy(), counts.numpy()
    assert np.array_equal(gc, want_c)
    for i in range(B):
        for lvl in range(L):
            c = want_c[i, lvl]
            assert np.array_equal(gs[i, lvl, :c], want_s[i, lvl, :c]) and np.isneginf(gs[i, lvl, c:]).all()
            np.testing.assert_allclose(gb[i, lvl, :c], want_b[i, lvl, :c], rtol=0, atol=2e-3)
    # post-NMS selection + ground truth
    keep = (rng.random((B, L, K)) < 0.3)
    keep &= np.arange(K)[None, None, :] < gc[:, :, None]
    keep[2] = False                                                               # an image that keeps nothing
    gt = rng.uniform(0, 100, (B, 4, 4)).astype(np.float32)
    gcnt = np.array([2, 0, 3], np.int32)
    pmax, post = 40 + 4, 40
    props, pcount = ctx.to_device(np.full((B, pmax, 4), np.nan, np.float32)), ctx.empty((B,), np.int32)
    dk, dg, dgc = ctx.to_device(keep.astype(np.uint8)), ctx.to_device(gt), ctx.to_device(gcnt)       # (kept alive across the launch)
    check(lib.rfi_op_proposals_select(ctx.handle, _P(boxes), _P(scores), _P(dk), B, L, K, post, _P(dg), 4, _P(dgc), pmax, _P(props), _P(pcount)))
    gp, gpc = props.numpy(), pcount.numpy()
    for i in range(B):
        fs = np.where(keep[i], gs[i], -np.inf).astype(np.float32).reshape(-1)
        sel = _topk_desc_stable(fs[None], post)[0][:min(int(keep[i].sum()), post)]
        want = np.concatenate([gb[i].reshape(-1, 4)[sel], gt[i, :gcnt[i]]])
        assert gpc[i] == len(want)
        assert np.array_equal(gp[i, :len(want)], want), i


def test_roi_sampler_compaction_and_levels():
    from oracle.mask_rcnn_ref import roi_levels, sample_order
    from rfi_toolbox_amd._lib import check, lib
    from rfi_toolbox_amd.models import MaskRCNN
    ctx = _ctx()
    rng = np.random.default_rng(3)
    B, pmax, gmax, batch, max_pos = 4, 60, 4, 32, 8
    pcount = np.array([60, 10, 45, 33], np.int32)
    p0 = rng.uniform(0, 100, (B, pmax, 2)).astype(np.float32)
    props = np.concatenate([p0, p0 + rng.uniform(2, 110, (B, pmax, 2)).astype(np.float32)], 2)
    labels = rng.choice([0, 1], size=(B, pmax), p=[0.7, 0.3]).astype(np.int8)
    labels[1] = 0                                                               # no foreground in image 1
    for i in range(B):
        labels[i, pcount[i]:] = -2
    matched = np.where(labels == 1, rng.integers(0, gmax, (B, pmax)), -1).astype(np.int32)
    targets = np.where((labels == 1)[..., None], rng.standard_normal((B, pmax, 4)), 0).astype(np.float32)
    gt_labels = rng.integers(1, 5, (B, gmax)).astype(np.int32)
    gbase = np.array([0, 3, 3, 7], np.int32)
    seed, step = 12345, 4
    hold = []

    def d(a):                                    # device copies stay referenced until the test ends
        hold.append(ctx.to_device(np.ascontiguousarray(a)))
        return hold[-1]
    sel, nsel, npos = ctx.empty((B, batch), np.int32), ctx.empty((B,), np.int32), ctx.empty((B,), np.int32)
    check(lib.rfi_op_roi_sample(ctx.handle, _P(d(labels)), _P(d(pcount)), B, pmax, batch, max_pos, seed, step, 2, _P(sel), _P(nsel), _P(npos)))
    R_, Rm = B * batch, B * max_pos
    o = {k: ctx.empty(sh, dt) for k, sh, dt in (("rois", (R_, 5), np.float32), ("cls", (R_,), np.int32), ("tgt", (R_, 4), np.float32),
                                                  ("gt", (R_,), np.int32), ("lvl", (R_,), np.int32), ("istart", (B + 1,), np.int32),
                                                  ("rfg", (Rm, 5), np.float32), ("rgt", (Rm, 5), np.float32), ("lfg", (Rm,), np.int32),
                                                  ("fstart", (B + 1,), np.int32), ("counts", (2,), np.int32))}
    t1, t2, t3 = MaskRCNN._level_thresholds(128)
    check(lib.rfi_op_roi_compact(ctx.handle, _P(sel), _P(nsel), _P(npos), B, batch, pmax, _P(d(props)), _P(d(matched)), _P(d(targets)),
                                 _P(d(gt_labels)), gmax, _P(d(gbase)), t1, t2, t3, _P(o["rois"]), _P(o["cls"]), _P(o["tgt"]), _P(o["gt"]),
                                 _P(o["lvl"]), _P(o["istart"]), _P(o["rfg"]), _P(o["rgt"]), _P(o["lfg"]), _P(o["fstart"]), _P(o["counts"])))
    rois, cls, tgt, rgt, fgr, fgg = [], [], [], [], [], []
    istart, fstart = [0], [0]
    for i in range(B):
        lab = labels[i, :pcount[i]]
        pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
        np_ = min(len(pos), max_pos)
        pos, neg = sample_order(pos, i, 2, seed, step)[:np_], sample_order(neg, i, 3, seed, step)[:batch - np_]
        keep = np.concatenate([pos, neg]).astype(int)
        r = np.concatenate([np.full((len(keep), 1), i, np.float32), props[i, keep]], 1)
        rois.append(r)
        c = np.zeros(len(keep), np.int32)
        c[:np_] = gt_labels[i, matched[i, pos]]
        cls.append(c); tgt.append(targets[i, keep]); rgt.append(np.where(np.arange(len(keep)) < np_, matched[i, keep], -1))
        fgr.append(r[:np_])
        fgg.append(np.concatenate([(gbase[i] + matched[i, pos])[:, None].astype(np.float32), props[i, pos]], 1))
        istart.append(istart[-1] + len(keep)); fstart.append(fstart[-1] + np_)
    rois, R, Rf = np.concatenate(rois), istart[-1], fstart[-1]
    assert list(o["counts"].numpy()) == [R, Rf] and list(o["istart"].numpy()) == istart and list(o["fstart"].numpy()) == fstart
    assert np.array_equal(o["rois"].numpy()[:R], rois) and np.array_equal(o["cls"].numpy()[:R], np.concatenate(cls))
    assert np.array_equal(o["tgt"].numpy()[:R], np.concatenate(tgt)) and np.array_equal(o["gt"].numpy()[:R], np.concatenate(rgt))
    assert np.array_equal(o["lvl"].numpy()[:R], roi_levels(rois[:, 1:], 128)) and len(set(o["lvl"].numpy()[:R])) >= 3
    assert np.array_equal(o["rfg"].numpy()[:Rf], np.concatenate(fgr)) and np.array_equal(o["rgt"].numpy()[:Rf], np.concatenate(fgg))
    assert np.array_equal(o["lfg"].numpy()[:Rf], roi_levels(np.concatenate(fgr)[:, 1:], 128))


def test_multi_level_roi_align_both_ways_against_the_single_level_oracle():
    from oracle import detection_ref
    from rfi_toolbox_amd._lib import check, lib
    ctx = _ctx()
    rng = np.random.default_rng(4)
    N, H0, W0, Cc = 3, 32, 32, 8
    feats = [rng.standard_normal((N, H0 >> k, W0 >> k, Cc)).astype(np.float32) for k in range(4)]
    rois, lv = [], []
    for n in range(N):                                                           # image-major RoI list, levels mixed inside an image
        for _ in range(7):
            x1, y1 = rng.uniform(-4, 100, 2)
            rois.append([n, x1, y1, x1 + rng.uniform(2, 90), y1 + rng.uniform(2, 90)])
            lv.append(int(rng.integers(0, 4)))
    rois, lv = np.asarray(rois, np.float32), np.asarray(lv, np.int32)
    R = len(rois)
    istart = np.array([0, 7, 14, 21], np.int32)
    df = [ctx.to_device(f) for f in feats]
    maps = (C.c_void_p * 4)(*[f.ptr for f in df])
    dr, dl, dc = ctx.to_device(rois), ctx.to_device(lv), ctx.to_device(np.array([R - 1, 0], np.int32))     # (one row fewer than the launch covers)
    for res in (7, 14):
        out = ctx.to_device(np.full((R, res, res, Cc), 7.0, np.float32))
        check(lib.rfi_op_roi_align_ml(ctx.handle, maps, N, H0, W0, Cc, 0.25, _P(dr), _P(dl), _P(dc), R, res, res, 2, _P(out)))
        got = out.numpy()
        for r in range(R - 1):
            want = detection_ref.roi_align(feats[lv[r]], rois[r:r + 1], 0.25 / (1 << lv[r]), (res, res), 2, False)[0]
            np.testing.assert_allclose(got[r], want, rtol=1e-5, atol=1e-5)
        assert (got[R - 1] == 7.0).all()                                         # beyond the device-side count: untouched
        # backward: ADDS into the level maps what detection_ref's adjoint gives per level
        dout = rng.standard_normal((R, res, res, Cc)).astype(np.float32)
        base = [rng.standard_normal(f.shape).astype(np.float32) for f in feats]
        dd = [ctx.to_device(a) for a in base]
        dmaps = (C.c_void_p * 4)(*[a.ptr for a in dd])
        ddo, dis = ctx.to_device(dout), ctx.to_device(istart)
        check(lib.rfi_op_roi_align_ml_backward(ctx.handle, dmaps, N, H0, W0, Cc, 0.25, _P(ddo), _P(dr), _P(dl), _P(dis), R, res, res, 2))
        for k in range(4):
            idx = np.flatnonzero(lv == k)
            want = base[k] + (detection_ref.roi_align_backward(dout[idx], feats[k].shape, rois[idx], 0.25 / (1 << k), (res, res), 2, False) if len(idx) else 0)
            np.testing.assert_allclose(dd[k].numpy(), want, rtol=2e-5, atol=2e-5)
