"""``MaskRCNN`` -- the detector API over the Mask R-CNN pieces (BASELINE.json configs[3]).

Not in the reference (it contains no detector; torchvision is absent): a builder-defined assembly of the published
pipeline (He et al. 2017) from the models and kernels of this package -- ``ResNet50FPN`` backbone, one ``RPNHead`` shared by
the five pyramid levels (``accumulate_gradients``), ``anchor_match`` / ``rpn_loss`` / ``decode_boxes`` / ``nms``,
multi-level ``roi_align``, ``BoxHead`` with ``fastrcnn_loss``, ``MaskHead``.  Every contraction, loss and gather runs on
the GPU.  ``train_step`` keeps every feature map, RoI feature, activation and gradient in HBM AND does the box bookkeeping
between the stages there too (round 4; ``csrc/detect_sample.hip``): the two samplers (counter-based: element i of image b
draws Philox4x32-10(counter (i, b, stream, step), key seed) and a class keeps its smallest draws), the per-level top-k of the
objectness scores, decode + clip, per-level NMS, the post-NMS selection, matching, the compact RoI lists and their pyramid
levels, multi-level RoIAlign both ways in one launch each.  What crosses PCIe inside a step: the ground-truth boxes going up,
two integers (RoI and foreground counts, read back under the RPN head's backward pass) and the loss scalars coming down.
``predict`` is the plain host-array form.  Conventions where implementations differ: box-coder weights 1 in both stages,
four anchors per pixel (aspect ratios 0.5, 1, 2 and a 1.5x square), level assignment
``k = clip(floor(4 + log2(sqrt(area) / (image_size / 2))), 2, 5)`` evaluated as three area comparisons.

    det = MaskRCNN(num_classes=2)
    losses = det.train_step(images_nhwc, [{"boxes": (g, 4), "labels": (g,), "masks": (g, H, W)}, ...])
    out = det.predict(images_nhwc)          # per image: boxes, scores, labels, masks (full-size bool), rfi_mask (union)
"""
from __future__ import annotations

import ctypes as C
import math

import os

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


_RPN_ALL_LATE = os.environ.get("RFI_RPN_ALL_LATE") is not None     # A/B runs: every level's RPN training pass behind the proposals

class MaskRCNN:
    def __init__(self, num_classes=2, in_channels=3, base_width=64, fpn_channels=256, representation_size=1024, *, device=None,
                 seed=None):
        self.num_classes, self.F = int(num_classes), int(fpn_channels)
        self.backbone = ResNet50FPN(in_channels, base_width, fpn_channels, device=device)
        self.rpn = RPNHead(fpn_channels, 4, 1, device=device)
        self.box = BoxHead(fpn_channels, 7, representation_size, num_classes, device=device)
        self.mask = MaskHead(fpn_channels, 1, 4, device=device)
        self.seed = int(np.random.SeedSequence(seed).generate_state(2, np.uint32).view(np.uint64)[0]) if seed is None else int(seed)
        self.sample_step = 0              # the samplers' step counter (advanced by train_step)
        self.keep_trace = False           # train_step downloads its discrete decisions into last_trace (tests; costs a few syncs)
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

    @staticmethod
    def _level_thresholds(size):
        half = np.float32(size) / np.float32(2.0)
        return tuple(float(np.float32(c * half) * np.float32(c * half)) for c in (0.5, 1.0, 2.0))

    def _levels(self, boxes, size):
        """Index into P2..P5: clip(floor(4 + log2(sqrt(area) / (size / 2))), 2, 5) - 2 as three comparisons of the float32
        area (no log2 / sqrt whose last bit could differ between host and device)."""
        boxes = np.asarray(boxes, np.float32)
        area = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), np.float32(1e-6))
        t1, t2, t3 = (np.float32(t) for t in self._level_thresholds(size))
        return (area >= t1).astype(int) + (area >= t2).astype(int) + (area >= t3).astype(int)

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

    # ---- one optimisation step: everything between the input batch and the loss scalars happens in HBM
    def _buffers(self, n, h, w, gmax):
        key = (n, h, w)
        b = getattr(self, "_buf", None)
        if getattr(self, "_buf_key", None) == key and b.gcap >= gmax:
            return b
        ctx, F = self.backbone.ctx, self.F
        b = type("Buffers", (), {})()
        b.gcap = max(8, 2 * gmax)
        b.x = ctx.empty((n, h, w, self.backbone.in_channels), np.float32)
        shapes = [(n, h // s, w // s, F) for s in _STRIDES]
        b.shapes = shapes
        b.feats = [ctx.empty(sh, np.float32) for sh in shapes]
        b.dfe = [ctx.empty(sh, np.float32) for sh in shapes]
        b.pf = (C.c_void_p * 5)(*[f.ptr for f in b.feats])
        b.pdf = (C.c_void_p * 5)(*[f.ptr for f in b.dfe])
        b.pf4 = (C.c_void_p * 4)(*[f.ptr for f in b.feats[:4]])
        b.pdf4 = (C.c_void_p * 4)(*[f.ptr for f in b.dfe[:4]])
        b.rpn_out = [ctx.empty((sh[0], sh[1], sh[2], 20), np.float32) for sh in shapes]
        b.rpn_dout = [ctx.empty((sh[0], sh[1], sh[2], 20), np.float32) for sh in shapes]
        b.rpn_lab = [ctx.empty((sh[0] * sh[1] * sh[2] * 4,), np.int8) for sh in shapes]
        b.rpn_tgt = [ctx.empty((sh[0] * sh[1] * sh[2] * 4, 4), np.float32) for sh in shapes]
        b.p_lab = (C.c_void_p * 5)(*[a.ptr for a in b.rpn_lab])
        b.p_tgt = (C.c_void_p * 5)(*[a.ptr for a in b.rpn_tgt])
        # anchors of the five levels (one image), concatenated; the matcher's and the samplers' tensors
        anchors = self._anchors(h, w)
        b.acount = [len(a) for a in anchors]
        b.aoff = np.concatenate([[0], np.cumsum(b.acount)]).astype(np.int32)
        b.A = int(b.aoff[-1])
        b.anchors = ctx.to_device(np.concatenate(anchors).astype(np.float32))
        pow2 = lambda v: 1 << max(1, int(v - 1).bit_length())  # noqa: E731
        b.gt, b.gt_count = ctx.empty((n, b.gcap, 4), np.float32), ctx.empty((n,), np.int32)
        b.gt_labels, b.gt_base = ctx.empty((n, b.gcap), np.int32), ctx.empty((n,), np.int32)
        b.best_ws = ctx.empty((n, b.gcap), np.float32)
        b.m_lab, b.m_idx, b.m_tgt = ctx.empty((n, b.A), np.int8), ctx.empty((n, b.A), np.int32), ctx.empty((n, b.A, 4), np.float32)
        b.stride_a = pow2(b.A)
        b.keys_a = ctx.empty((n, b.stride_a), np.uint64)
        b.n_sampled = ctx.empty((4,), np.int32)
        b.stride_l = [max(2, pow2(c)) for c in b.acount]
        b.keys_l = [ctx.empty((n, st), np.uint64) for st in b.stride_l]
        K, L = self.pre_nms, 5
        b.cand_boxes, b.cand_scores = ctx.empty((n, L, K, 4), np.float32), ctx.empty((n, L, K), np.float32)
        b.cand_counts, b.keep = ctx.empty((n, L), np.int32), ctx.empty((n, L, K), np.uint8)
        b.pmax = self.post_nms + b.gcap
        b.props, b.pcount = ctx.empty((n, b.pmax, 4), np.float32), ctx.empty((n,), np.int32)
        b.r_lab, b.r_idx, b.r_tgt = ctx.empty((n, b.pmax), np.int8), ctx.empty((n, b.pmax), np.int32), ctx.empty((n, b.pmax, 4), np.float32)
        b.sel, b.nsel, b.npos = ctx.empty((n, self.roi_batch), np.int32), ctx.empty((n,), np.int32), ctx.empty((n,), np.int32)
        R, Rm = n * self.roi_batch, n * (self.roi_batch // 4)
        b.rois, b.rois_m, b.rois_g = ctx.empty((R, 5), np.float32), ctx.empty((Rm, 5), np.float32), ctx.empty((Rm, 5), np.float32)
        b.roi_gt, b.roi_lvl, b.lvl_m = ctx.empty((R,), np.int32), ctx.empty((R,), np.int32), ctx.empty((Rm,), np.int32)
        b.img_start, b.fg_start, b.counts = ctx.empty((n + 1,), np.int32), ctx.empty((n + 1,), np.int32), ctx.empty((4,), np.int32)
        b.roi7, b.roi7_grad = ctx.empty((R, 7, 7, F), np.float32), ctx.empty((R, 7, 7, F), np.float32)
        k1 = self.num_classes
        b.box_out, b.box_dout = ctx.empty((R, 5 * k1), np.float32), ctx.empty((R, 5 * k1), np.float32)
        b.box_lab, b.box_tgt = ctx.empty((R,), np.int32), ctx.empty((R, 4), np.float32)
        b.roi14, b.roi14_grad = ctx.empty((Rm, 14, 14, F), np.float32), ctx.empty((Rm, 14, 14, F), np.float32)
        b.mask_t = ctx.empty((Rm, 28, 28), np.uint8)
        b.masks, b.masks_n, b.masks_of = None, 0, None
        ws = int(lib.rfi_op_rpn_loss_ws_bytes())
        b.rpn_ws = [ctx.empty((ws,), np.uint8) for _ in shapes]
        b.box_ws = ctx.empty((ws,), np.uint8)
        b.rpn_loss2, b.box_loss2 = ctx.empty((len(shapes), 2), np.float32), ctx.empty((2,), np.float32)
        self._buf_key, self._buf = key, b
        return b

    def train_step(self, images, targets, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, masks_resident=False):
        """One training step.  masks_resident=True: the caller promises that `targets` (the SAME list object as in the
        previous step) still holds the same instance masks, so the copy already in HBM is used; otherwise the masks are
        uploaded every step (a list refilled in place must not train the mask branch on last step's masks)."""
        from ..runtime import DeviceArray
        dev_in = isinstance(images, DeviceArray)                 # images already in HBM (float32 NHWC): no copy
        x = images if dev_in else np.ascontiguousarray(np.asarray(images, np.float32))
        n, h, w, _ = x.shape
        ctx, F, k1 = self.backbone.ctx, self.F, self.num_classes
        H = ctx.handle
        P = lambda d: C.c_void_p(d.ptr)  # noqa: E731
        for m in (self.rpn, self.box, self.mask):
            m.train()
        gts = [np.asarray(t["boxes"], np.float32).reshape(-1, 4) for t in targets]
        gcount = np.asarray([len(g) for g in gts], np.int32)
        b = self._buffers(n, h, w, int(gcount.max()) if n else 0)
        if dev_in:
            b.x_in = x
        else:
            b.x.copy_from(x)
            b.x_in = b.x
        G = b.gcap
        # ---- up: the ground truth (boxes, class labels, counts) -- a few hundred bytes per image
        gt = np.zeros((n, G, 4), np.float32)
        gl = np.zeros((n, G), np.int32)
        for i, (g, t) in enumerate(zip(gts, targets)):
            gt[i, :len(g)] = g
            gl[i, :len(g)] = np.asarray(t["labels"], np.int32).reshape(-1)
        gbase = np.concatenate([[0], np.cumsum(gcount)]).astype(np.int32)
        b.gt.copy_from(gt); b.gt_labels.copy_from(gl); b.gt_count.copy_from(gcount); b.gt_base.copy_from(gbase[:n])
        b.n_sampled.zero_()
        seed, step = self.seed & 0xFFFFFFFFFFFFFFFF, self.sample_step & 0xFFFFFFFF
        self.sample_step += 1
        # ---- RPN targets: matcher (IoU 0.7 / 0.3, low-quality matches) + sampler (256 per image, at most half positive)
        check(lib.rfi_op_anchor_match_batched_ws(H, P(b.anchors), b.A, 0, None, P(b.gt), n, G, P(b.gt_count), 0.7, 0.3, 1, P(b.best_ws),
                                                 P(b.m_lab), P(b.m_idx), P(b.m_tgt)))
        check(lib.rfi_op_sample_keys(H, P(b.m_lab), n, b.A, None, seed, step, 0, P(b.keys_a), b.stride_a))
        check(lib.rfi_op_segsort_u64(H, P(b.keys_a), n, b.stride_a))
        check(lib.rfi_op_rpn_sample_apply(H, P(b.keys_a), n, b.A, b.stride_a, self.rpn_batch, self.rpn_batch // 2, P(b.m_lab), P(b.m_tgt),
                                          5, b.aoff.ctypes.data_as(C.c_void_p), b.p_lab, b.p_tgt, P(b.n_sampled)))
        # ---- backbone, RPN head on every level
        check(lib.rfi_backbone_forward(self.backbone._h, P(b.x_in), DEVICE, n, h, w, b.pf, DEVICE))

        def rpn_backward(lvl):
            """Loss (global normaliser on the device) and backward pass of the shared head at one level; its activations of this
            level must be the ones the head holds."""
            _, hl, wl, _ = b.shapes[lvl]
            check(lib.rfi_op_rpn_loss_devcount(H, P(b.rpn_out[lvl]), n * hl * wl, 4, P(b.rpn_lab[lvl]), P(b.rpn_tgt[lvl]), P(b.n_sampled),
                                               1.0 / 9, P(b.rpn_dout[lvl]), P(b.rpn_ws[lvl]), C.c_void_p(b.rpn_loss2.ptr + 8 * lvl)))
            check(lib.rfi_model_backward_dlogits(self.rpn._h, P(b.feats[lvl]), DEVICE, P(b.rpn_dout[lvl]), DEVICE, n, hl, wl))
            self.rpn.accumulate_gradients("add")
            check(lib.rfi_model_input_grad(self.rpn._h, P(b.dfe[lvl]), DEVICE))          # the first term of d loss / d P_l

        # The head keeps the activations of ONE pass.  Level 0 (three quarters of the head's work) trains at once, while its
        # activations are there -- its targets were sampled above; the other levels' passes come back after the proposals,
        # behind the read-back of the RoI counts (their forward passes are repeated then: a sixteenth of level 0's each)
        self.rpn.accumulate_gradients("begin")
        for lvl in (0, 1, 2, 3, 4):
            _, hl, wl, _ = b.shapes[lvl]
            check(lib.rfi_model_forward_nhwc(self.rpn._h, P(b.feats[lvl]), DEVICE, n, hl, wl, P(b.rpn_out[lvl]), DEVICE))
            if lvl == 0 and not _RPN_ALL_LATE:
                rpn_backward(0)
        # ---- proposals: top pre_nms per level -> decode + clip -> per-level NMS -> best post_nms + ground truth
        K = self.pre_nms
        for lvl in range(5):
            _, hl, wl, _ = b.shapes[lvl]
            check(lib.rfi_op_topk_keys(H, P(b.rpn_out[lvl]), n, hl * wl, 4, P(b.keys_l[lvl]), b.stride_l[lvl]))
            check(lib.rfi_op_segsort_u64(H, P(b.keys_l[lvl]), n, b.stride_l[lvl]))
            check(lib.rfi_op_topk_decode(H, P(b.keys_l[lvl]), n, b.stride_l[lvl], hl * wl, 4, K, P(b.rpn_out[lvl]),
                                         C.c_void_p(b.anchors.ptr + int(b.aoff[lvl]) * 16), float(h), float(w), 1e-2, P(b.cand_boxes),
                                         P(b.cand_scores), P(b.cand_counts), 5, lvl))
        check(lib.rfi_op_nms_batched(H, P(b.cand_boxes), P(b.cand_counts), n * 5, K, float(self.rpn_nms), P(b.keep)))
        check(lib.rfi_op_proposals_select(H, P(b.cand_boxes), P(b.cand_scores), P(b.keep), n, 5, K, self.post_nms, P(b.gt), G,
                                          P(b.gt_count), b.pmax, P(b.props), P(b.pcount)))
        # ---- RoI targets: matcher at 0.5, sampler (128 per image, at most a quarter foreground), compact lists + levels
        check(lib.rfi_op_anchor_match_batched_ws(H, P(b.props), b.pmax, b.pmax, P(b.pcount), P(b.gt), n, G, P(b.gt_count), 0.5, 0.5, 0,
                                                 P(b.best_ws), P(b.r_lab), P(b.r_idx), P(b.r_tgt)))
        check(lib.rfi_op_roi_sample(H, P(b.r_lab), P(b.pcount), n, b.pmax, self.roi_batch, self.roi_batch // 4, seed, step, 2, P(b.sel),
                                    P(b.nsel), P(b.npos)))
        t1, t2, t3 = self._level_thresholds(max(h, w))
        check(lib.rfi_op_roi_compact(H, P(b.sel), P(b.nsel), P(b.npos), n, self.roi_batch, b.pmax, P(b.props), P(b.r_idx), P(b.r_tgt),
                                     P(b.gt_labels), G, P(b.gt_base), t1, t2, t3, P(b.rois), P(b.box_lab), P(b.box_tgt), P(b.roi_gt),
                                     P(b.roi_lvl), P(b.img_start), P(b.rois_m), P(b.rois_g), P(b.lvl_m), P(b.fg_start), P(b.counts)))
        check(lib.rfi_readback_begin(H, P(b.counts), 8))         # (R, Rf) come down while the RPN head's backward passes run
        # ---- the rest of the RPN head's own training work (level 4 first: its forward pass was the last one above)
        for lvl in ((4, 0, 1, 2, 3) if _RPN_ALL_LATE else (4, 1, 2, 3)):
            _, hl, wl, _ = b.shapes[lvl]
            if lvl != 4:
                check(lib.rfi_model_forward_nhwc(self.rpn._h, P(b.feats[lvl]), DEVICE, n, hl, wl, P(b.rpn_out[lvl]), DEVICE))
            rpn_backward(lvl)
        self.rpn.accumulate_gradients("end")
        cnt2 = (C.c_int32 * 2)()
        check(lib.rfi_readback_end(H, cnt2, 8))
        R, Rf = int(cnt2[0]), int(cnt2[1])
        losses = {"loss_objectness": 0.0, "loss_rpn_box_reg": 0.0, "loss_classifier": 0.0, "loss_box_reg": 0.0, "loss_mask": 0.0}
        h0, w0 = h // 4, w // 4
        # ---- box head on the sampled RoIs (multi-level RoIAlign: one launch each way)
        if R:
            check(lib.rfi_op_roi_align_ml(H, b.pf4, n, h0, w0, F, 0.25, P(b.rois), P(b.roi_lvl), P(b.counts), R, 7, 7, 2, P(b.roi7)))
            check(lib.rfi_model_forward_nhwc(self.box._h, P(b.roi7), DEVICE, R, 1, 1, P(b.box_out), DEVICE))
            check(lib.rfi_op_fastrcnn_loss_dev(H, P(b.box_out), R, k1, P(b.box_lab), P(b.box_tgt), 1.0 / 9, P(b.box_dout), P(b.box_ws),
                                               P(b.box_loss2)))
            check(lib.rfi_model_backward_dlogits(self.box._h, P(b.roi7), DEVICE, P(b.box_dout), DEVICE, R, 1, 1))
            check(lib.rfi_model_input_grad(self.box._h, P(b.roi7_grad), DEVICE))
            check(lib.rfi_op_roi_align_ml_backward(H, b.pdf4, n, h0, w0, F, 0.25, P(b.roi7_grad), P(b.rois), P(b.roi_lvl), P(b.img_start), R,
                                                   7, 7, 2))
        # ---- mask branch on the foreground RoIs: targets = the matched ground-truth mask, RoIAligned to 28 x 28 at 0.5
        if Rf:
            if b.masks is None or b.masks_n < gbase[-1]:
                b.masks, b.masks_n, b.masks_of = ctx.empty((int(gbase[-1]), h, w), np.uint8), int(gbase[-1]), None
            if not (masks_resident and b.masks_of is targets):   # (resident only on the caller's word AND for the same list object)
                self._upload(b.masks, np.concatenate([np.asarray(t["masks"], np.uint8).reshape(-1, h, w) for t in targets]))
                b.masks_of = targets
            check(lib.rfi_op_roi_align_ml(H, b.pf4, n, h0, w0, F, 0.25, P(b.rois_m), P(b.lvl_m), C.c_void_p(b.counts.ptr + 4), Rf, 14, 14, 2,
                                          P(b.roi14)))
            check(lib.rfi_op_mask_targets(H, P(b.masks), int(gbase[-1]), h, w, P(b.rois_g), Rf, 28, 28, 2, P(b.mask_t)))
            check(lib.rfi_train_forward_backward(self.mask._h, P(b.roi14), DEVICE, P(b.mask_t), DEVICE, Rf, 14, 14, None))
            check(lib.rfi_model_input_grad(self.mask._h, P(b.roi14_grad), DEVICE))
            check(lib.rfi_op_roi_align_ml_backward(H, b.pdf4, n, h0, w0, F, 0.25, P(b.roi14_grad), P(b.rois_m), P(b.lvl_m), P(b.fg_start), Rf,
                                                   14, 14, 2))
        check(lib.rfi_backbone_backward(self.backbone._h, P(b.x_in), DEVICE, n, h, w, b.pdf, DEVICE))
        norms = {}
        for name, m in zip(("backbone", "rpn", "box", "mask"), self.models()):
            if m is self.mask and not Rf:
                # no foreground RoI on THIS rank.  Alone (grad_sync == 1) the mask head skips its step; in a data-parallel job
                # the other ranks enter the all-reduce of its gradients, so this rank must too -- with zero gradients -- and
                # then apply the same averaged update (a rank that skipped would hang the collective or leave the replicas
                # different)
                if self.grad_sync <= 1:
                    continue
                m.accumulate_gradients("begin")
                m.accumulate_gradients("end")                    # grads = 0
            if m is self.box and not R:
                continue
            if self.grad_sync > 1:
                m.allreduce_gradients()
            norms[name] = m.apply_gradients(lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                            grad_scale=1.0 / max(self.grad_sync, 1))
        # ---- down: the loss scalars
        l2 = b.rpn_loss2.numpy()
        losses["loss_objectness"], losses["loss_rpn_box_reg"] = float(l2[:, 0].sum(dtype=np.float32)), float(l2[:, 1].sum(dtype=np.float32))
        if R:
            bl = b.box_loss2.numpy()
            losses["loss_classifier"], losses["loss_box_reg"] = float(bl[0]), float(bl[1])
        if Rf:
            losses["loss_mask"] = float(self.mask.last_loss()[0])
        losses["loss"] = float(sum(losses.values()))
        self.last_trace = {"grad_norms": norms, "num_rois": R, "num_foreground": Rf}
        if self.keep_trace:
            # the discrete decisions of the step (tests replay them through oracle/mask_rcnn_ref.py)
            pc = b.pcount.numpy()
            pr = b.props.numpy()
            self.last_trace.update({
                "rpn_labels": np.concatenate([b.rpn_lab[l].numpy().reshape(n, -1) for l in range(5)], 1),
                "rpn_targets": np.concatenate([b.rpn_tgt[l].numpy().reshape(n, -1, 4) for l in range(5)], 1),
                "proposals": [pr[i, :pc[i]].copy() for i in range(n)], "rois": b.rois.numpy()[:R], "roi_labels": b.box_lab.numpy()[:R],
                "roi_targets": b.box_tgt.numpy()[:R], "roi_gt": b.roi_gt.numpy()[:R], "roi_levels": b.roi_lvl.numpy()[:R],
                "num_sampled": int(b.n_sampled.numpy()[0])})
        return losses

    @staticmethod
    def _upload(dev, arr):
        """Host array -> the leading bytes of a (larger) device buffer."""
        arr = np.ascontiguousarray(arr, dtype=dev.dtype)
        if arr.nbytes > dev.nbytes:
            raise ValueError("device buffer too small")
        if arr.nbytes:
            check(lib.rfi_memcpy(dev.ctx.handle, C.c_void_p(dev.ptr), DEVICE, arr.ctypes.data_as(C.c_void_p), HOST, arr.nbytes))
