"""``MaskRCNN`` -- the detector API over the Mask R-CNN pieces (BASELINE.json configs[3]).

Not in the reference (it contains no detector; torchvision is absent): a builder-defined assembly of the published
pipeline (He et al. 2017) from the models and kernels of this package -- ``ResNet50FPN`` backbone, one ``RPNHead`` shared by
the five pyramid levels (``accumulate_gradients``), ``anchor_match`` / ``rpn_loss`` / ``decode_boxes`` / ``nms``,
multi-level ``roi_align``, ``BoxHead`` with ``fastrcnn_loss``, ``MaskHead``.  Every contraction, loss and gather runs on
the GPU; this class is the host-side bookkeeping between them (anchor grids, the random samplers, level assignment, score
thresholds, mask pasting), with host round trips between the stages -- it defines the API, it is not the fast path
(``tools/bench_maskrcnn_lite.py`` times the device-resident chain).  Conventions where implementations differ: box-coder
weights 1 in both stages, four anchors per pixel (aspect ratios 0.5, 1, 2 and a 1.5x square), level assignment
``k = floor(k0 + log2(sqrt(area) / s0))`` with ``(k0, s0) = (4, image_size / 2)``.

    det = MaskRCNN(num_classes=2)
    losses = det.train_step(images_nhwc, [{"boxes": (g, 4), "labels": (g,), "masks": (g, H, W)}, ...])
    out = det.predict(images_nhwc)          # per image: boxes, scores, labels, masks (full-size bool), rfi_mask (union)
"""
from __future__ import annotations

import math

import numpy as np

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
        best overall (+ ``extra`` boxes: the ground truth during training)."""
        props = []
        for i in range(n):
            boxes, scores = [], []
            for lvl, (o, a) in enumerate(zip(rpn_out, anchors)):
                oi = o[i].reshape(-1, 20)
                sc, dl = oi[:, :4].reshape(-1), oi[:, 4:].reshape(-1, 4)
                top = np.argsort(-sc, kind="stable")[:self.pre_nms]
                b = ops.decode_boxes(a[top], dl[top], image_size=(h, w))        # one delta row per selected anchor
                ok = ((b[:, 2] - b[:, 0]) >= 1e-2) & ((b[:, 3] - b[:, 1]) >= 1e-2)
                b, s_ = b[ok], sc[top][ok]
                keep = ops.nms(b, s_, self.rpn_nms) if len(b) else np.zeros(0, np.int64)
                boxes.append(b[keep]); scores.append(s_[keep])
            b, s_ = np.concatenate(boxes), np.concatenate(scores)
            b = b[np.argsort(-s_, kind="stable")[:self.post_nms]]
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
    def train_step(self, images, targets, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0):
        x = np.ascontiguousarray(np.asarray(images, np.float32))
        n, h, w, _ = x.shape
        for m in (self.rpn, self.box, self.mask):
            m.train()
        feats = self.backbone.forward_features(x)
        shapes = [f.shape for f in feats]
        dfe = [np.zeros(s, np.float32) for s in shapes]
        anchors = self._anchors(h, w)
        all_anchors = np.concatenate(anchors)
        # RPN targets: match, then sample rpn_batch anchors per image (at most half positive)
        labels = np.empty((n, len(all_anchors)), np.int8)
        tgts = np.empty((n, len(all_anchors), 4), np.float32)
        for i in range(n):
            lab, _, tg = ops.anchor_match(all_anchors, targets[i]["boxes"])
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.rpn_batch // 2)
            lab[self.rng.permutation(pos)[npos:]] = -1
            lab[self.rng.permutation(neg)[self.rpn_batch - npos:]] = -1
            labels[i], tgts[i] = lab, tg
        n_sampled = max(int((labels >= 0).sum()), 1)
        losses = {"loss_objectness": 0.0, "loss_rpn_box_reg": 0.0}
        rpn_out, off = [], 0
        self.rpn.accumulate_gradients("begin")
        for lvl, f in enumerate(feats):
            cnt = len(anchors[lvl])
            o = self.rpn.forward_nhwc(f)
            rpn_out.append(o)
            lo, lb, dout = ops.rpn_loss(o.reshape(-1, 20), labels[:, off:off + cnt].reshape(-1), tgts[:, off:off + cnt].reshape(-1, 4), 4,
                                        num_sampled=n_sampled)
            self.rpn.backward(f, dout)
            self.rpn.accumulate_gradients("add")
            dfe[lvl] += self.rpn.input_grad(f.shape)
            losses["loss_objectness"] += lo
            losses["loss_rpn_box_reg"] += lb
            off += cnt
        self.rpn.accumulate_gradients("end")
        # RoI heads: proposals (+ ground truth) matched at IoU 0.5, roi_batch per image with at most a quarter foreground
        props = self._proposals(rpn_out, anchors, n, h, w, extra=[t["boxes"] for t in targets])
        rois, rlab, rtgt, rgt = [], [], [], []
        for i, p in enumerate(props):
            g = np.asarray(targets[i]["boxes"], np.float32).reshape(-1, 4)
            lab, midx, tg = ops.anchor_match(p, g, 0.5, 0.5, False)
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.roi_batch // 4)
            pos, neg = self.rng.permutation(pos)[:npos], self.rng.permutation(neg)[:self.roi_batch - npos]
            keep = np.concatenate([pos, neg])
            cls = np.zeros(len(keep), np.int32)
            cls[:npos] = np.asarray(targets[i]["labels"], np.int32).reshape(-1)[midx[pos]]
            rois.append(np.concatenate([np.full((len(keep), 1), i, np.float32), p[keep]], 1))
            rlab.append(cls); rtgt.append(tg[keep]); rgt.append(np.where(np.arange(len(keep)) < npos, midx[keep], -1))
        rois, rlab, rtgt, rgt = np.concatenate(rois), np.concatenate(rlab), np.concatenate(rtgt), np.concatenate(rgt)
        rf, lv = self._roi_align(feats, rois, 7, max(h, w))
        head = self.box.forward_rois(rf)
        lc, lr_, dout = ops.fastrcnn_loss(head, rlab, rtgt)
        self.box.backward(rf, dout)
        self._roi_align_backward(self.box.input_grad(rf.shape), shapes, rois, lv, dfe)
        losses["loss_classifier"], losses["loss_box_reg"] = lc, lr_
        # mask branch on the foreground RoIs: targets = the matched ground-truth mask, RoIAligned to 28 x 28
        fg = np.flatnonzero(rlab > 0)
        losses["loss_mask"] = 0.0
        if len(fg):
            mf, mlv = self._roi_align(feats, rois[fg], 14, max(h, w))
            mt = np.zeros((len(fg), 28, 28), np.uint8)
            for j, r in enumerate(fg):
                gm = np.asarray(targets[int(rois[r, 0])]["masks"], np.float32)[rgt[r]]
                src = np.repeat(gm[None, :, :, None], 4, 3)                   # (1, H, W, 4): the kernels want C % 4 == 0
                roi = np.concatenate([[0.0], rois[r, 1:]]).astype(np.float32)[None]
                mt[j] = ops.roi_align(src, roi, 1.0, (28, 28), 2, False)[0, :, :, 0] >= 0.5
            losses["loss_mask"] = self.mask.forward_backward(mf, mt)
            self._roi_align_backward(self.mask.input_grad(mf.shape), shapes, rois[fg], mlv, dfe)
        self.backbone.backward(x, dfe)
        for m in self.models():
            if m is self.mask and not len(fg):
                continue
            m.apply_gradients(lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        losses["loss"] = float(sum(losses.values()))
        return losses
