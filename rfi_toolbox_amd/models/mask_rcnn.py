"""``MaskRCNN`` -- the detector API over the Mask R-CNN pieces (BASELINE.json configs[3]).

Not in the reference (it contains no detector; torchvision is absent): a builder-defined assembly of the published
pipeline (He et al. 2017) from the models and kernels of this package -- ``ResNet50FPN`` backbone, one ``RPNHead`` shared by
the five pyramid levels (``accumulate_gradients``), ``anchor_match`` / ``rpn_loss`` / ``decode_boxes`` / ``nms``,
multi-level ``roi_align``, ``BoxHead`` with ``fastrcnn_loss``, ``MaskHead``.  Every contraction, loss and gather runs on
the GPU; this class is the host-side bookkeeping between them (anchor grids, the random samplers, level assignment, score
thresholds, mask pasting).  ``train_step`` keeps every feature map, RoI feature, activation and gradient in HBM and calls
the C-ABI on device pointers; what crosses to the host is the box bookkeeping (RPN head outputs for the top-k, labels of the
batched matcher, the sampled index sets, five loss scalars).  ``predict`` is the plain host-array form.  Conventions where implementations differ: box-coder
weights 1 in both stages, four anchors per pixel (aspect ratios 0.5, 1, 2 and a 1.5x square), level assignment
``k = floor(k0 + log2(sqrt(area) / s0))`` with ``(k0, s0) = (4, image_size / 2)``.

    det = MaskRCNN(num_classes=2)
    losses = det.train_step(images_nhwc, [{"boxes": (g, 4), "labels": (g,), "masks": (g, H, W)}, ...])
    out = det.predict(images_nhwc)          # per image: boxes, scores, labels, masks (full-size bool), rfi_mask (union)
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .._lib import DEVICE, HOST, check, lib
from . import detection_ops as ops
from .backbone import ResNet50FPN
from .box_head import BoxHead
from .mask_head import MaskHead
from .rpn_head import RPNHead

_STRIDES = (4, 8, 16, 32, 64)


def _level_anchors(h, w, stride, size):
    """(h w A, 4) anchors of one level, pixel-major / anchor-minor (the order of the RPN head's output)."""
    shapes = [(size * math.sqrt(r), size / math.sqrt(r)) for r in (0.5, 1.0, 2.0)] + [(1.5 * size, 1.5 * size)]     # (w, h)
    ys, xs = np.meshgrid((np.arange(h) + 0.5) * stride, (np.arange(w) + 0.5) * stride, indexing="ij")
    out = np.empty((h, w, 4, 4), np.float32)
    for a, (aw, ah) in enumerate(shapes):
        out[..., a, 0], out[..., a, 1] = xs - aw / 2, ys - ah / 2
        out[..., a, 2], out[..., a, 3] = xs + aw / 2, ys + ah / 2
    return out.reshape(-1, 4)


def _topk_desc_stable(scores, k):
    """Row-wise indices of the k largest float32 scores in descending order, ties by ascending index -- what
    ``np.argsort(-scores, kind="stable")[:, :k]`` returns, through a partial selection on exact 64-bit integer keys
    (order-preserving image of the float32 in the high word, the index in the low word) instead of a full sort."""
    sc = np.ascontiguousarray(scores, np.float32)
    n, m = sc.shape
    bits = ((-sc) + np.float32(0.0)).view(np.uint32)                   # ascending -score = descending score (+ 0: -0.0 and 0.0 tie)
    key = np.where(bits & 0x80000000, ~bits, bits | 0x80000000).astype(np.uint64)          # monotone map float32 -> uint32
    key = (key << np.uint64(32)) | np.arange(m, dtype=np.uint64)[None, :]
    if k >= m:
        return np.argsort(key, axis=1)
    part = np.argpartition(key, k - 1, axis=1)[:, :k]
    rows = np.arange(n)[:, None]
    return part[rows, np.argsort(key[rows, part], axis=1)]


def _paste(prob, box, h, w):
    """28 x 28 mask probabilities -> boolean mask of the (h, w) image inside ``box`` (bilinear, threshold 0.5)."""
    x1, y1, x2, y2 = [float(v) for v in box]
    out = np.zeros((h, w), bool)
    ix1, iy1, ix2, iy2 = max(int(math.floor(x1)), 0), max(int(math.floor(y1)), 0), min(int(math.ceil(x2)), w), min(int(math.ceil(y2)), h)
    if ix2 <= ix1 or iy2 <= iy1:
        return out
    m = prob.shape[0]
    gx = (np.arange(ix1, ix2) + 0.5 - x1) / max(x2 - x1, 1e-6) * m - 0.5
    gy = (np.arange(iy1, iy2) + 0.5 - y1) / max(y2 - y1, 1e-6) * m - 0.5
    x0, y0 = np.clip(np.floor(gx).astype(int), 0, m - 1), np.clip(np.floor(gy).astype(int), 0, m - 1)
    x1i, y1i = np.clip(x0 + 1, 0, m - 1), np.clip(y0 + 1, 0, m - 1)
    fx, fy = np.clip(gx - x0, 0, 1)[None, :], np.clip(gy - y0, 0, 1)[:, None]
    v = (prob[y0][:, x0] * (1 - fx) + prob[y0][:, x1i] * fx) * (1 - fy) + (prob[y1i][:, x0] * (1 - fx) + prob[y1i][:, x1i] * fx) * fy
    out[iy1:iy2, ix1:ix2] = v > 0.5
    return out


class MaskRCNN:
    def __init__(self, num_classes=2, in_channels=3, base_width=64, fpn_channels=256, representation_size=1024, *, device=None,
                 seed=None):
        self.num_classes, self.F = int(num_classes), int(fpn_channels)
        self.backbone = ResNet50FPN(in_channels, base_width, fpn_channels, device=device)
        self.rpn = RPNHead(fpn_channels, 4, 1, device=device)
        self.box = BoxHead(fpn_channels, 7, representation_size, num_classes, device=device)
        self.mask = MaskHead(fpn_channels, 1, 4, device=device)
        self.rng = np.random.default_rng(seed)
        self.pre_nms, self.post_nms, self.rpn_nms, self.rpn_batch, self.roi_batch = 200, 100, 0.7, 256, 128
        self.score_thresh, self.det_nms, self.max_det = 0.05, 0.5, 20
        self.grad_sync = 0                # data parallel over this many ranks (> 1): all-reduce (mean) of the four gradient sets

    def models(self):
        return (self.backbone, self.rpn, self.box, self.mask)

    def set_compute_dtype(self, dtype):
        for m in self.models():
            m.set_compute_dtype(dtype)
        return self

    # ---- geometry
    def _anchors(self, h, w):
        return [_level_anchors(h // s, w // s, s, 2.0 * s) for s in _STRIDES]

    def _levels(self, boxes, size):
        area = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), 1e-6)
        k = np.floor(4 + np.log2(np.sqrt(area) / (size / 2.0) + 1e-9))
        return np.clip(k, 2, 5).astype(int) - 2                     # index into P2..P5

    def _roi_align(self, feats, rois, out, size):
        """Multi-level RoIAlign: every RoI on its pyramid level; rois (R, 5) = (image, x1, y1, x2, y2)."""
        res = np.zeros((len(rois), out, out, self.F), np.float32)
        lv = self._levels(rois[:, 1:], size)
        for k in range(4):
            idx = np.flatnonzero(lv == k)
            if len(idx):
                res[idx] = ops.roi_align(feats[k], rois[idx], 1.0 / _STRIDES[k], (out, out), 2, False)
        return res, lv

    def _roi_align_backward(self, dres, feats_shapes, rois, lv, dfe):
        for k in range(4):
            idx = np.flatnonzero(lv == k)
            if len(idx):
                dfe[k] += ops.roi_align_backward(dres[idx], feats_shapes[k], rois[idx], 1.0 / _STRIDES[k], 2, False)

    def _proposals(self, rpn_out, anchors, n, h, w, extra=None):
        """Per image: top ``pre_nms`` boxes per level by objectness, decoded and clipped, per-level NMS, the ``post_nms``
        best overall (+ ``extra`` boxes: the ground truth during training).  One decode launch per level and ONE NMS
        launch for all (image, level) sets."""
        K, L = self.pre_nms, len(anchors)
        sets_b = np.zeros((n, L, K, 4), np.float32)
        sets_s = np.full((n, L, K), -np.inf, np.float32)
        counts = np.zeros((n, L), np.int32)
        rows = np.arange(n)[:, None]
        for lvl, (o, a) in enumerate(zip(rpn_out, anchors)):
            oi = np.asarray(o).reshape(n, -1, 20)
            sc, dl = oi[:, :, :4].reshape(n, -1), oi[:, :, 4:].reshape(n, -1, 4)
            top = _topk_desc_stable(sc, K)                                                  # (n, k) anchors by descending score
            k = top.shape[1]
            bx = ops.decode_boxes(a[top].reshape(-1, 4), dl[rows, top].reshape(-1, 4), image_size=(h, w)).reshape(n, k, 4)
            ss = sc[rows, top]
            ok = ((bx[..., 2] - bx[..., 0]) >= 1e-2) & ((bx[..., 3] - bx[..., 1]) >= 1e-2)
            first = np.argsort(~ok, axis=1, kind="stable")                                  # degenerate boxes to the back, order kept
            sets_b[:, lvl, :k] = np.take_along_axis(bx, first[..., None], 1)
            sets_s[:, lvl, :k] = np.take_along_axis(np.where(ok, ss, -np.inf).astype(np.float32), first, 1)
            counts[:, lvl] = ok.sum(1)
        keep = ops.nms_batched(sets_b.reshape(-1, K, 4), counts.reshape(-1), self.rpn_nms).reshape(n, L, K)
        # the post_nms best of an image over its levels: stable descending order of the level-major list of kept boxes
        flat_s = np.where(keep, sets_s, -np.inf).astype(np.float32).reshape(n, L * K)
        flat_b = sets_b.reshape(n, L * K, 4)
        sel = _topk_desc_stable(flat_s, self.post_nms)
        nsel = np.minimum(keep.sum((1, 2)), self.post_nms)
        props = []
        for i in range(n):
            b = flat_b[i, sel[i, :nsel[i]]]
            if extra is not None and len(extra[i]):
                b = np.concatenate([b, np.asarray(extra[i], np.float32).reshape(-1, 4)])
            props.append(b.astype(np.float32))
        return props

    # ---- inference
    def predict(self, images):
        x = np.ascontiguousarray(np.asarray(images, np.float32))
        n, h, w, _ = x.shape
        for m in (self.rpn, self.box, self.mask):
            m.eval()
        feats = self.backbone.forward_features(x)
        anchors = self._anchors(h, w)
        rpn_out = [self.rpn.forward_nhwc(f) for f in feats]
        props = self._proposals(rpn_out, anchors, n, h, w)
        rois = np.concatenate([np.concatenate([np.full((len(p), 1), i, np.float32), p], 1) for i, p in enumerate(props)])
        out = []
        k1 = self.num_classes
        if len(rois):
            rf, _ = self._roi_align(feats, rois, 7, max(h, w))
            head = self.box.forward_rois(rf)
            z = head[:, :k1] - head[:, :k1].max(1, keepdims=True)
            prob = np.exp(z) / np.exp(z).sum(1, keepdims=True)
        for i in range(n):
            sel = np.flatnonzero(rois[:, 0] == i) if len(rois) else np.zeros(0, int)
            boxes, scores, labels = [], [], []
            for c in range(1, k1):
                if not len(sel):
                    break
                b = ops.decode_boxes(rois[sel, 1:], head[sel, k1 + 4 * c:k1 + 4 * c + 4], image_size=(h, w))
                s_ = prob[sel, c]
                ok = (s_ > self.score_thresh) & ((b[:, 2] - b[:, 0]) >= 1e-2) & ((b[:, 3] - b[:, 1]) >= 1e-2)
                b, s_ = b[ok], s_[ok]
                keep = ops.nms(b, s_, self.det_nms) if len(b) else np.zeros(0, np.int64)
                boxes.append(b[keep]); scores.append(s_[keep]); labels.append(np.full(len(keep), c, np.int64))
            boxes = np.concatenate(boxes) if boxes else np.zeros((0, 4), np.float32)
            scores = np.concatenate(scores) if scores else np.zeros(0, np.float32)
            labels = np.concatenate(labels) if labels else np.zeros(0, np.int64)
            top = np.argsort(-scores, kind="stable")[:self.max_det]
            boxes, scores, labels = boxes[top], scores[top], labels[top]
            masks = np.zeros((len(boxes), h, w), bool)
            if len(boxes):
                dr = np.concatenate([np.full((len(boxes), 1), i, np.float32), boxes], 1)
                mf, _ = self._roi_align(feats, dr, 14, max(h, w))
                logit = self.mask.forward_nhwc(mf)[..., 0]
                pm = 1.0 / (1.0 + np.exp(-logit))
                for j in range(len(boxes)):
                    masks[j] = _paste(pm[j], boxes[j], h, w)
            out.append({"boxes": boxes, "scores": scores, "labels": labels, "masks": masks,
                        "rfi_mask": masks.any(0) if len(masks) else np.zeros((h, w), bool)})
        return out

    # ---- one optimisation step
    # Every feature map, RoI feature, head activation and gradient stays in HBM (device buffers cached per input shape,
    # C-ABI calls on device pointers); what crosses to the host is the small integer / box bookkeeping: RPN head outputs
    # for top-k + NMS, anchor / RoI labels from the matcher, the sampled index sets, five loss scalars.
    def _buffers(self, n, h, w):
        key = (n, h, w)
        if getattr(self, "_buf_key", None) == key:
            return self._buf
        ctx, F = self.backbone.ctx, self.F
        b = type("Buffers", (), {})()
        b.x = ctx.empty((n, h, w, self.backbone.in_channels), np.float32)
        shapes = [(n, h // s, w // s, F) for s in _STRIDES]
        b.shapes = shapes
        b.feats = [ctx.empty(sh, np.float32) for sh in shapes]
        b.dfe = [ctx.empty(sh, np.float32) for sh in shapes]
        b.tmp = [ctx.empty(sh, np.float32) for sh in shapes[:4]]
        b.pf = (C.c_void_p * 5)(*[f.ptr for f in b.feats])
        b.pdf = (C.c_void_p * 5)(*[f.ptr for f in b.dfe])
        b.rpn_out = [ctx.empty((sh[0], sh[1], sh[2], 20), np.float32) for sh in shapes]
        b.rpn_dout = [ctx.empty((sh[0], sh[1], sh[2], 20), np.float32) for sh in shapes]
        b.rpn_lab = [ctx.empty((sh[0] * sh[1] * sh[2] * 4,), np.int8) for sh in shapes]
        b.rpn_tgt = [ctx.empty((sh[0] * sh[1] * sh[2] * 4, 4), np.float32) for sh in shapes]
        R, Rm = n * self.roi_batch, n * (self.roi_batch // 4)
        b.rois, b.rois_m, b.rois_g = ctx.empty((R, 5), np.float32), ctx.empty((Rm, 5), np.float32), ctx.empty((Rm, 5), np.float32)
        b.roi7, b.roi7_grad = ctx.empty((R, 7, 7, F), np.float32), ctx.empty((R, 7, 7, F), np.float32)
        k1 = self.num_classes
        b.box_out, b.box_dout = ctx.empty((R, 5 * k1), np.float32), ctx.empty((R, 5 * k1), np.float32)
        b.box_lab, b.box_tgt = ctx.empty((R,), np.int32), ctx.empty((R, 4), np.float32)
        b.roi14, b.roi14_grad = ctx.empty((Rm, 14, 14, F), np.float32), ctx.empty((Rm, 14, 14, F), np.float32)
        b.mask_t = ctx.empty((Rm, 28, 28), np.uint8)
        b.masks, b.masks_n, b.masks_of = None, 0, None
        b.rpn_ws = [ctx.empty((int(lib.rfi_op_rpn_loss_ws_bytes()),), np.uint8) for _ in shapes]
        b.rpn_loss2 = ctx.empty((len(shapes), 2), np.float32)
        self._buf_key, self._buf = key, b
        return b

    def _rpn_match(self, all_anchors, targets):
        """Matcher of the RPN targets (IoU 0.7 / 0.3, low-quality matches; one launch for the batch).
        -> labels int8 (n, A) in {1, 0, -1}, regression targets (n, A, 4)."""
        labels, _, tgts = ops.anchor_match_batched(all_anchors, [t["boxes"] for t in targets])
        return labels, tgts

    def _rpn_sample(self, labels):
        """The random sampler on the matcher's labels (in place): rpn_batch anchors per image, at most half positive;
        the others get -1 = not sampled."""
        for i in range(len(labels)):
            lab = labels[i]
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.rpn_batch // 2)
            lab[self.rng.permutation(pos)[npos:]] = -1
            lab[self.rng.permutation(neg)[self.rpn_batch - npos:]] = -1
        return labels

    def _rpn_targets(self, all_anchors, targets):
        labels, tgts = self._rpn_match(all_anchors, targets)
        return self._rpn_sample(labels), tgts

    def _roi_match(self, props, targets):
        """Proposals (+ ground truth) against the ground truth at IoU 0.5 (one launch for the batch)."""
        pmax = max(len(p) for p in props)
        pb = np.zeros((len(props), pmax, 4), np.float32)
        for i, p in enumerate(props):
            pb[i, :len(p)] = p
        return ops.anchor_match_batched(pb, [t["boxes"] for t in targets], 0.5, 0.5, False, anchor_counts=[len(p) for p in props])

    def _roi_sample(self, props, targets, match):
        """The RoI sampler: roi_batch per image with at most a quarter foreground.  -> rois (R, 5), class labels (R,),
        regression targets (R, 4), matched ground-truth index (R,) (-1: background)."""
        labs, midxs, tgs = match
        rois, rlab, rtgt, rgt = [], [], [], []
        for i, p in enumerate(props):
            lab, midx, tg = labs[i, :len(p)], midxs[i, :len(p)], tgs[i, :len(p)]
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.roi_batch // 4)
            pos, neg = self.rng.permutation(pos)[:npos], self.rng.permutation(neg)[:self.roi_batch - npos]
            keep = np.concatenate([pos, neg])
            cls = np.zeros(len(keep), np.int32)
            cls[:npos] = np.asarray(targets[i]["labels"], np.int32).reshape(-1)[midx[pos]]
            rois.append(np.concatenate([np.full((len(keep), 1), i, np.float32), p[keep]], 1))
            rlab.append(cls); rtgt.append(tg[keep]); rgt.append(np.where(np.arange(len(keep)) < npos, midx[keep], -1))
        return np.concatenate(rois), np.concatenate(rlab), np.concatenate(rtgt).astype(np.float32), np.concatenate(rgt)

    def _sample_rois(self, props, targets):
        return self._roi_sample(props, targets, self._roi_match(props, targets))

    def _roi_align_dev(self, b, rois_dev, lv, out_dev, res, backward=False, grad_dev=None):
        """Multi-level RoIAlign on device buffers; the RoIs are SORTED by level, so level k is one contiguous slice of the
        RoI list and of the output.  backward: RoI-feature gradients -> added to the level's feature gradient."""
        ctx, F = self.backbone.ctx, self.F
        off = 0
        for k in range(4):
            cnt = int((lv == k).sum())
            if cnt:
                n, hk, wk, _ = b.shapes[k]
                rp = C.c_void_p(rois_dev.ptr + off * 20)
                fp = C.c_void_p((grad_dev if backward else out_dev).ptr + off * res * res * F * 4)
                if backward:          # (gather form: the level's RoIs are in image order -- the level sort is stable; no atomics)
                    check(lib.rfi_op_roi_align_backward_sorted(ctx.handle, fp, n, hk, wk, F, rp, cnt, 1.0 / _STRIDES[k], res, res, 2, 0,
                                                               C.c_void_p(b.tmp[k].ptr)))
                    check(lib.rfi_op_add_inplace(ctx.handle, C.c_void_p(b.dfe[k].ptr), C.c_void_p(b.tmp[k].ptr), n * hk * wk * F))
                else:
                    check(lib.rfi_op_roi_align(ctx.handle, C.c_void_p(b.feats[k].ptr), n, hk, wk, F, rp, cnt, 1.0 / _STRIDES[k],
                                               res, res, 2, 0, fp))
            off += cnt

    def train_step(self, images, targets, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, masks_resident=False):
        """One training step.  masks_resident=True: the caller promises that `targets` (the SAME list object as in the
        previous step) still holds the same instance masks, so the copy already in HBM is used; otherwise the masks are
        uploaded every step (a list refilled in place must not train the mask branch on last step's masks)."""
        from ..runtime import DeviceArray
        dev_in = isinstance(images, DeviceArray)                 # images already in HBM (float32 NHWC): no copy
        x = images if dev_in else np.ascontiguousarray(np.asarray(images, np.float32))
        n, h, w, _ = x.shape
        ctx, F, k1 = self.backbone.ctx, self.F, self.num_classes
        P = lambda d: C.c_void_p(d.ptr)  # noqa: E731
        for m in (self.rpn, self.box, self.mask):
            m.train()
        b = self._buffers(n, h, w)
        if dev_in:
            b.x_in = x
        else:
            b.x.copy_from(x)
            b.x_in = b.x
        # The order of the calls below overlaps the host's bookkeeping with the GPU: the matcher first (the GPU is idle
        # anyway), then the backbone is enqueued and the host runs the anchor sampler under it; the RPN head's loss and
        # backward passes are enqueued (no host round trip: rfi_op_rpn_loss_dev) BEHIND the proposal / matching launches, so
        # they run while the host samples and sorts the RoIs.
        anchors = self._anchors(h, w)
        all_anchors = np.concatenate(anchors)
        labels, tgts = self._rpn_match(all_anchors, targets)
        check(lib.rfi_backbone_forward(self.backbone._h, P(b.x_in), DEVICE, n, h, w, b.pf, DEVICE))
        labels = self._rpn_sample(labels)
        n_sampled = max(int((labels >= 0).sum()), 1)
        losses = {"loss_objectness": 0.0, "loss_rpn_box_reg": 0.0}
        rpn_out, off = [], 0
        for lvl in range(5):
            cnt = len(anchors[lvl])
            _, hl, wl, _ = b.shapes[lvl]
            b.rpn_lab[lvl].copy_from(labels[:, off:off + cnt].reshape(-1))
            b.rpn_tgt[lvl].copy_from(tgts[:, off:off + cnt].reshape(-1, 4))
            check(lib.rfi_model_forward_nhwc(self.rpn._h, P(b.feats[lvl]), DEVICE, n, hl, wl, P(b.rpn_out[lvl]), DEVICE))
            off += cnt
        rpn_out = [b.rpn_out[lvl].numpy() for lvl in range(5)]
        # RoI heads' inputs: proposals and their matches (GPU launches + host top-k), THEN the RPN head's own training work
        props = self._proposals(rpn_out, anchors, n, h, w, extra=[t["boxes"] for t in targets])
        match = self._roi_match(props, targets)
        self.rpn.accumulate_gradients("begin")
        for lvl in (4, 0, 1, 2, 3):                              # (level 4 first: its forward pass was the last one above)
            _, hl, wl, _ = b.shapes[lvl]
            check(lib.rfi_op_rpn_loss_dev(ctx.handle, P(b.rpn_out[lvl]), n * hl * wl, 4, P(b.rpn_lab[lvl]), P(b.rpn_tgt[lvl]), n_sampled,
                                          1.0 / 9, P(b.rpn_dout[lvl]), P(b.rpn_ws[lvl]), C.c_void_p(b.rpn_loss2.ptr + 8 * lvl)))
            # (forward again: the head keeps the activations of ONE pass, and later passes overwrote this level's)
            if lvl != 4:
                check(lib.rfi_model_forward_nhwc(self.rpn._h, P(b.feats[lvl]), DEVICE, n, hl, wl, P(b.rpn_out[lvl]), DEVICE))
            check(lib.rfi_model_backward_dlogits(self.rpn._h, P(b.feats[lvl]), DEVICE, P(b.rpn_dout[lvl]), DEVICE, n, hl, wl))
            self.rpn.accumulate_gradients("add")
            check(lib.rfi_model_input_grad(self.rpn._h, P(b.dfe[lvl]), DEVICE))          # the first term of d loss / d P_l
        self.rpn.accumulate_gradients("end")
        rois, rlab, rtgt, rgt = self._roi_sample(props, targets, match)                     # (host; the GPU is in the RPN passes)
        lv = self._levels(rois[:, 1:], max(h, w))
        order = np.argsort(lv, kind="stable")                    # level-major: a level is one slice of every RoI tensor
        rois, rlab, rtgt, rgt, lv = rois[order], rlab[order], rtgt[order], rgt[order], lv[order]
        R = len(rois)
        self._upload(b.rois, rois); self._upload(b.box_lab, rlab.astype(np.int32)); self._upload(b.box_tgt, rtgt)
        self._roi_align_dev(b, b.rois, lv, b.roi7, 7)
        check(lib.rfi_model_forward_nhwc(self.box._h, P(b.roi7), DEVICE, R, 1, 1, P(b.box_out), DEVICE))
        lc, lr_ = C.c_float(), C.c_float()
        check(lib.rfi_op_fastrcnn_loss(ctx.handle, P(b.box_out), R, k1, P(b.box_lab), P(b.box_tgt), 1.0 / 9, P(b.box_dout),
                                       C.byref(lc), C.byref(lr_)))
        check(lib.rfi_model_backward_dlogits(self.box._h, P(b.roi7), DEVICE, P(b.box_dout), DEVICE, R, 1, 1))
        check(lib.rfi_model_input_grad(self.box._h, P(b.roi7_grad), DEVICE))
        self._roi_align_dev(b, b.rois, lv, None, 7, backward=True, grad_dev=b.roi7_grad)
        losses["loss_classifier"], losses["loss_box_reg"] = lc.value, lr_.value
        # mask branch on the foreground RoIs: targets = the matched ground-truth mask, RoIAligned to 28 x 28 at 0.5
        fg = np.flatnonzero(rlab > 0)
        losses["loss_mask"] = 0.0
        if len(fg):
            gcount = [len(t["boxes"]) for t in targets]
            gbase = np.concatenate([[0], np.cumsum(gcount)])
            if b.masks is None or b.masks_n < gbase[-1]:
                b.masks, b.masks_n, b.masks_of = ctx.empty((int(gbase[-1]), h, w), np.uint8), int(gbase[-1]), None
            if not (masks_resident and b.masks_of is targets):   # (resident only on the caller's word AND for the same list object)
                self._upload(b.masks, np.concatenate([np.asarray(t["masks"], np.uint8).reshape(-1, h, w) for t in targets]))
                b.masks_of = targets
            rm = rois[fg]
            rg = np.concatenate([(gbase[rm[:, 0].astype(int)] + rgt[fg])[:, None].astype(np.float32), rm[:, 1:]], 1)
            self._upload(b.rois_m, rm); self._upload(b.rois_g, rg)
            Rf = len(fg)
            self._roi_align_dev(b, b.rois_m, lv[fg], b.roi14, 14)
            check(lib.rfi_op_mask_targets(ctx.handle, P(b.masks), int(gbase[-1]), h, w, P(b.rois_g), Rf, 28, 28, 2, P(b.mask_t)))
            lm = C.c_float()
            check(lib.rfi_train_forward_backward(self.mask._h, P(b.roi14), DEVICE, P(b.mask_t), DEVICE, Rf, 14, 14, C.byref(lm)))
            check(lib.rfi_model_input_grad(self.mask._h, P(b.roi14_grad), DEVICE))
            self._roi_align_dev(b, b.rois_m, lv[fg], None, 14, backward=True, grad_dev=b.roi14_grad)
            losses["loss_mask"] = lm.value
        check(lib.rfi_backbone_backward(self.backbone._h, P(b.x_in), DEVICE, n, h, w, b.pdf, DEVICE))
        norms = {}
        for name, m in zip(("backbone", "rpn", "box", "mask"), self.models()):
            if m is self.mask and not len(fg):
                # no foreground RoI on THIS rank.  Alone (grad_sync == 1) the mask head skips its step; in a data-parallel job
                # the other ranks enter the all-reduce of its gradients, so this rank must too -- with zero gradients -- and
                # then apply the same averaged update (a rank that skipped would hang the collective or leave the replicas
                # different)
                if self.grad_sync <= 1:
                    continue
                m.accumulate_gradients("begin")
                m.accumulate_gradients("end")                    # grads = 0
            if self.grad_sync > 1:
                m.allreduce_gradients()
            norms[name] = m.apply_gradients(lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                            grad_scale=1.0 / max(self.grad_sync, 1))
        l2 = b.rpn_loss2.numpy()
        losses["loss_objectness"], losses["loss_rpn_box_reg"] = float(l2[:, 0].sum(dtype=np.float32)), float(l2[:, 1].sum(dtype=np.float32))
        losses["loss"] = float(sum(losses.values()))
        # the discrete decisions and gradient norms of the step (tests replay them through oracle/mask_rcnn_ref.py)
        self.last_trace = {"rpn_labels": labels, "rpn_targets": tgts, "proposals": props, "rois": rois, "roi_labels": rlab,
                           "roi_targets": rtgt, "roi_gt": rgt, "roi_levels": lv, "grad_norms": norms}
        return losses

    @staticmethod
    def _upload(dev, arr):
        """Host array -> the leading bytes of a (larger) device buffer."""
        arr = np.ascontiguousarray(arr, dtype=dev.dtype)
        if arr.nbytes > dev.nbytes:
            raise ValueError("device buffer too small")
        if arr.nbytes:
            check(lib.rfi_memcpy(dev.ctx.handle, C.c_void_p(dev.ptr), DEVICE, arr.ctypes.data_as(C.c_void_p), HOST, arr.nbytes))
