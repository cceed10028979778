"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the ASSEMBLED Mask R-CNN training step (BASELINE.json
configs[3]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
(rfi_toolbox_amd) never does.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox contains no detector (placeholder strings only, README.md:90,381,
docs/API.md:180) and torchvision is absent from this image.  This file restates the published training step (Ren et al.
2015; He et al. 2017; Lin et al. 2017) with the conventions of rfi_toolbox_amd.models.MaskRCNN's docstring, from the
pieces the other oracle files hold:

    backbone_ref.ResNet50FPN                     -> [P2 .. P6]
    mask_head_ref.RPNHeadModule on every level   -> objectness + deltas; anchors: 4 per pixel (ratios 0.5, 1, 2 and a 1.5 x square
                                                    of size 2 x stride)
    detection_ref.anchor_match (0.7 / 0.3, low-quality) + sampler (256 per image, at most half positive)   -> RPN losses
    proposals: top 200 per level, decode + clip, drop boxes under 0.01, NMS 0.7 per level, best 100, + ground truth
    detection_ref.anchor_match (0.5 / 0.5) + sampler (128 per image, at most a quarter foreground)
    level k = clip(floor(4 + log2(sqrt(area) / (size / 2))), 2, 5);  RoIAlign 7 x 7 -> BoxHeadModule -> Fast R-CNN losses
    foreground RoIs: RoIAlign 14 x 14 -> MaskHeadModule -> BCE against the matched instance mask RoIAligned to 28 x 28 (>= 0.5)

everything in float32 torch / NumPy on the CPU, autograd for the gradients.  The random samplers are COUNTER BASED (round 4:
the device step samples in HBM, rfi_toolbox_amd/csrc/detect_sample.hip): candidate i of image b draws
r = Philox4x32-10(counter (i, b, stream, step), key seed).x and a class keeps its candidates of smallest (r, i); streams 0 / 1
= RPN positives / negatives, 2 / 3 = RoI foreground / background.  The step is a function of (weights, batch, seed, step).
``decisions`` replays the discrete choices of another run (labels, proposals, RoIs) so that continuous quantities can be
compared even where a threshold decision differs in the last bit.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import backbone_ref, detection_ref, mask_head_ref
from .synth_ref import philox4x32_10

STRIDES = (4, 8, 16, 32, 64)


def sample_order(idx, image, stream, seed, step):
    """The candidates ``idx`` of one class of image ``image`` in the order the counter-based sampler prefers them: ascending
    (Philox word, index)."""
    idx = np.asarray(idx, np.int64)
    if len(idx) == 0:
        return idx
    r = philox4x32_10(idx, np.full(len(idx), image), np.full(len(idx), stream), np.full(len(idx), step), int(seed) & 0xFFFFFFFF,
                      (int(seed) >> 32) & 0xFFFFFFFF)[0].astype(np.uint64)
    return idx[np.lexsort((idx, r))]


def level_anchors(h, w, stride, size):
    """(h w 4, 4) anchors of one level, pixel-major / anchor-minor: ratios 0.5, 1, 2 of area size^2 and a 1.5 x square."""
    shapes = [(size * math.sqrt(r), size / math.sqrt(r)) for r in (0.5, 1.0, 2.0)] + [(1.5 * size, 1.5 * size)]     # (w, h)
    out = np.empty((h, w, 4, 4), np.float32)
    for y in range(h):
        for x in range(w):
            cx, cy = (x + 0.5) * stride, (y + 0.5) * stride
            for a, (aw, ah) in enumerate(shapes):
                out[y, x, a] = (cx - aw / 2, cy - ah / 2, cx + aw / 2, cy + ah / 2)
    return out.reshape(-1, 4)


def level_thresholds(size):
    """Areas (float32) at which a box moves up a pyramid level: k = clip(floor(4 + log2(sqrt(area) / (size / 2))), 2, 5) written
    as comparisons (exact in float32 on any device: no log2, no sqrt)."""
    half = np.float32(size) / np.float32(2.0)
    return tuple(np.float32(c * half) * np.float32(c * half) for c in (0.5, 1.0, 2.0))


def roi_levels(boxes, size):
    boxes = np.asarray(boxes, np.float32)
    area = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), np.float32(1e-6))
    t1, t2, t3 = level_thresholds(size)
    return (area >= t1).astype(int) + (area >= t2).astype(int) + (area >= t3).astype(int)                # index into P2..P5


def roi_align_torch(feat, rois, scale, res, sampling_ratio=2):
    """Differentiable RoIAlign (rules of detection_ref.roi_align, not aligned): feat (N, C, H, W) tensor, rois (R, 5) array
    -> (R, C, res, res).  The bilinear corner indices and weights are built in NumPy, the gather + weighted sum in torch."""
    N, C, H, W = feat.shape
    R = len(rois)
    if R == 0:
        return feat.new_zeros((0, C, res, res))
    rois = np.asarray(rois, np.float32)
    g = sampling_ratio
    x1, y1 = rois[:, 1] * np.float32(scale), rois[:, 2] * np.float32(scale)
    rw = np.maximum(rois[:, 3] * np.float32(scale) - x1, np.float32(1.0))
    rh = np.maximum(rois[:, 4] * np.float32(scale) - y1, np.float32(1.0))
    bw, bh = rw / np.float32(res), rh / np.float32(res)
    p = np.arange(res, dtype=np.float32)
    s = (np.arange(g, dtype=np.float32) + np.float32(0.5)) / np.float32(g)
    ys = (y1[:, None, None] + p[None, :, None] * bh[:, None, None] + s[None, None, :] * bh[:, None, None]).reshape(R, res * g)
    xs = (x1[:, None, None] + p[None, :, None] * bw[:, None, None] + s[None, None, :] * bw[:, None, None]).reshape(R, res * g)

    def axis(v, L):
        ok = ~((v < -1.0) | (v > L))
        v = np.maximum(v, 0.0)
        lo = np.floor(v).astype(np.int64)
        top = lo >= L - 1
        lo = np.where(top, L - 1, lo)
        hi = np.where(top, L - 1, lo + 1)
        fr = np.where(top, 0.0, v - lo).astype(np.float32)
        return lo, hi, fr, ok

    y0, y1i, fy, oky = axis(ys, H)
    x0, x1i, fx, okx = axis(xs, W)
    n = torch.as_tensor(rois[:, 0].astype(np.int64))
    fp = feat.permute(0, 2, 3, 1)                                                # (N, H, W, C)

    def corner(yi, xi, wy, wx):
        wgt = torch.as_tensor((wy[:, :, None] * wx[:, None, :]) * (oky[:, :, None] & okx[:, None, :]))        # (R, Sy, Sx)
        v = fp[n[:, None, None], torch.as_tensor(yi)[:, :, None], torch.as_tensor(xi)[:, None, :]]              # (R, Sy, Sx, C)
        return v * wgt[..., None]

    acc = corner(y0, x0, 1 - fy, 1 - fx) + corner(y0, x1i, 1 - fy, fx) + corner(y1i, x0, fy, 1 - fx) + corner(y1i, x1i, fy, fx)
    acc = acc.reshape(R, res, g, res, g, C).sum((2, 4)) / float(g * g)
    return acc.permute(0, 3, 1, 2)


class MaskRCNNRef:
    def __init__(self, num_classes=2, in_channels=3, base_width=64, fpn_channels=256, representation_size=1024):
        self.num_classes, self.F = num_classes, fpn_channels
        self.backbone = backbone_ref.ResNet50FPN(in_channels, base_width, fpn_channels).eval()
        self.rpn = mask_head_ref.RPNHeadModule(fpn_channels, 4, 1)
        self.box = mask_head_ref.BoxHeadModule(fpn_channels, 7, representation_size, num_classes)
        self.mask = mask_head_ref.MaskHeadModule(fpn_channels, 1, 4)
        self.pre_nms, self.post_nms, self.rpn_nms, self.rpn_batch, self.roi_batch = 200, 100, 0.7, 256, 128

    def modules(self):
        return {"backbone": self.backbone, "rpn": self.rpn, "box": self.box, "mask": self.mask}

    def load(self, states):
        """states: {"backbone" | "rpn" | "box" | "mask": state_dict} (the key names of the torch modules)."""
        for k, m in self.modules().items():
            m.load_state_dict({n: torch.as_tensor(np.asarray(v)).to(torch.float32) for n, v in states[k].items()})
        return self

    # ---- the discrete parts
    def rpn_targets(self, all_anchors, targets, sampler):
        seed, step = sampler
        n = len(targets)
        labels = np.empty((n, len(all_anchors)), np.int8)
        tgts = np.empty((n, len(all_anchors), 4), np.float32)
        for i in range(n):
            lab, _, tg = detection_ref.anchor_match(all_anchors, np.asarray(targets[i]["boxes"], np.float32).reshape(-1, 4))
            lab = lab.astype(np.int8)
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.rpn_batch // 2)
            lab[sample_order(pos, i, 0, seed, step)[npos:]] = -1
            lab[sample_order(neg, i, 1, seed, step)[self.rpn_batch - npos:]] = -1
            labels[i], tgts[i] = lab, tg
        return labels, tgts

    def proposals(self, rpn_out, anchors, n, h, w, extra):
        props = []
        for i in range(n):
            boxes, scores = [], []
            for o, a in zip(rpn_out, anchors):
                oi = o[i].reshape(-1, 20)
                sc, dl = oi[:, :4].reshape(-1), oi[:, 4:].reshape(-1, 4)
                top = np.argsort(-sc, kind="stable")[:self.pre_nms]
                b = detection_ref.decode_boxes(a[top], dl[top], image_size=(h, w))
                ok = ((b[:, 2] - b[:, 0]) >= 1e-2) & ((b[:, 3] - b[:, 1]) >= 1e-2)
                b, s_ = b[ok], sc[top][ok]
                keep = detection_ref.nms(b, s_, self.rpn_nms) if len(b) else np.zeros(0, np.int64)
                boxes.append(b[keep]); scores.append(s_[keep])
            b, s_ = np.concatenate(boxes), np.concatenate(scores)
            b = b[np.argsort(-s_, kind="stable")[:self.post_nms]]
            if len(extra[i]):
                b = np.concatenate([b, np.asarray(extra[i], np.float32).reshape(-1, 4)])
            props.append(b.astype(np.float32))
        return props

    def sample_rois(self, props, targets, sampler):
        seed, step = sampler
        rois, rlab, rtgt, rgt = [], [], [], []
        for i, p in enumerate(props):
            g = np.asarray(targets[i]["boxes"], np.float32).reshape(-1, 4)
            lab, midx, tg = detection_ref.anchor_match(p, g, 0.5, 0.5, False)
            pos, neg = np.flatnonzero(lab == 1), np.flatnonzero(lab == 0)
            npos = min(len(pos), self.roi_batch // 4)
            pos, neg = sample_order(pos, i, 2, seed, step)[:npos], sample_order(neg, i, 3, seed, step)[:self.roi_batch - npos]
            keep = np.concatenate([pos, neg])
            cls = np.zeros(len(keep), np.int32)
            cls[:npos] = np.asarray(targets[i]["labels"], np.int32).reshape(-1)[midx[pos]]
            rois.append(np.concatenate([np.full((len(keep), 1), i, np.float32), p[keep]], 1))
            rlab.append(cls); rtgt.append(tg[keep]); rgt.append(np.where(np.arange(len(keep)) < npos, midx[keep], -1))
        rois, rlab, rtgt, rgt = np.concatenate(rois), np.concatenate(rlab), np.concatenate(rtgt).astype(np.float32), np.concatenate(rgt)
        return rois, rlab, rtgt, rgt

    # ---- the step: losses, gradient norms of the four parameter sets, the decisions taken
    def step(self, images_nhwc, targets, sampler=None, decisions=None, grads=True):
        """sampler = (seed, step) of the counter-based samplers (needed unless ``decisions`` replays every discrete choice)."""
        x = torch.as_tensor(np.asarray(images_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
        n, _, h, w = x.shape
        dec = decisions or {}
        for m in self.modules().values():
            m.zero_grad(set_to_none=True)
        feats = self.backbone(x)                                              # [P2 .. P6], NCHW
        anchors = [level_anchors(h // s, w // s, s, 2.0 * s) for s in STRIDES]
        all_anchors = np.concatenate(anchors)
        if "rpn_labels" in dec:
            labels, tgts = np.asarray(dec["rpn_labels"]), np.asarray(dec["rpn_targets"], np.float32)
        else:
            labels, tgts = self.rpn_targets(all_anchors, targets, sampler)
        n_sampled = max(int((labels >= 0).sum()), 1)
        l_obj, l_box, off, rpn_out = 0.0, 0.0, 0, []
        for lvl, f in enumerate(feats):
            cnt = len(anchors[lvl])
            cls, box = self.rpn(f)
            out = torch.cat([cls, box], 1).permute(0, 2, 3, 1)                # (N, H, W, 5 A)
            rpn_out.append(out.detach().numpy())
            P_, A = out.shape[0] * out.shape[1] * out.shape[2], 4
            o = out.reshape(P_, 5 * A)
            lab = torch.as_tensor(labels[:, off:off + cnt].reshape(P_, A).astype(np.int64))
            tgt = torch.as_tensor(tgts[:, off:off + cnt].reshape(P_, A, 4))
            xo, d = o[:, :A], o[:, A:].reshape(P_, A, 4)
            samp, pos = lab >= 0, lab > 0
            l_obj = l_obj + F.binary_cross_entropy_with_logits(xo[samp], pos[samp].to(o.dtype), reduction="sum") / n_sampled
            l_box = l_box + F.smooth_l1_loss(d[pos], tgt[pos], beta=1.0 / 9, reduction="sum") / n_sampled
            off += cnt
        extra = [np.asarray(t["boxes"], np.float32).reshape(-1, 4) for t in targets]
        props = [np.asarray(p, np.float32) for p in dec["proposals"]] if "proposals" in dec else self.proposals(rpn_out, anchors, n, h, w, extra)
        if "rois" in dec:
            rois, rlab, rtgt, rgt = (np.asarray(dec[k]) for k in ("rois", "roi_labels", "roi_targets", "roi_gt"))
            rois, rtgt = rois.astype(np.float32), rtgt.astype(np.float32)
        else:
            rois, rlab, rtgt, rgt = self.sample_rois(props, targets, sampler)
        lv = roi_levels(rois[:, 1:], max(h, w))

        def pooled(sel, res):
            out = feats[0].new_zeros((len(sel), self.F, res, res))
            for k in range(4):
                idx = np.flatnonzero(lv[sel] == k)
                if len(idx):
                    out[torch.as_tensor(idx)] = roi_align_torch(feats[k], rois[sel][idx], 1.0 / STRIDES[k], res)
            return out

        allr = np.arange(len(rois))
        cls, box = self.box(pooled(allr, 7))
        l_cls, l_reg = mask_head_ref.fastrcnn_loss_torch(cls, box, rlab, rtgt)
        fg = np.flatnonzero(rlab > 0)
        l_mask = torch.zeros(())
        mt = np.zeros((0, 28, 28), np.uint8)
        if len(fg):
            mt = np.zeros((len(fg), 28, 28), np.uint8)
            for j, r in enumerate(fg):
                gm = np.asarray(targets[int(rois[r, 0])]["masks"], np.float32)[rgt[r]]
                roi = np.concatenate([[0.0], rois[r, 1:]]).astype(np.float32)[None]
                mt[j] = detection_ref.roi_align(gm[None, :, :, None], roi, 1.0, (28, 28), 2, False)[0, :, :, 0] >= 0.5
            logits = self.mask(pooled(fg, 14))
            l_mask = mask_head_ref.mask_loss(logits[:, 0], torch.as_tensor(mt))
        total = l_obj + l_box + l_cls + l_reg + l_mask
        val = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)  # noqa: E731
        losses = {"loss_objectness": val(l_obj), "loss_rpn_box_reg": val(l_box), "loss_classifier": val(l_cls),
                  "loss_box_reg": val(l_reg), "loss_mask": val(l_mask), "loss": val(total)}
        norms = {}
        if grads:
            total.backward()
            for k, m in self.modules().items():
                sq = sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None)
                norms[k] = math.sqrt(sq)
        trace = {"rpn_labels": labels, "rpn_targets": tgts, "proposals": props, "rois": rois, "roi_labels": rlab, "roi_targets": rtgt,
                 "roi_gt": rgt, "roi_levels": lv, "mask_targets": mt, "rpn_out": rpn_out, "grad_norms": norms}
        return losses, trace
