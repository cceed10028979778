"""NumPy oracle of the detection building blocks (RoIAlign, FPN top-down merge).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox contains no detector (README.md:90 and docs/API.md:180 only
name a "detector"), and torchvision is absent from this image, so these functions restate the PUBLISHED algorithms
-- RoIAlign of He et al. 2017 with the sampling rules of the public torchvision.ops.roi_align, and the top-down
pathway of Lin et al. 2017 -- in plain float64 loops.  Feature maps are NHWC like every tensor of the path."""
import math

import numpy as np


def _bilinear(feat, y, x):
    """feat (H, W, C); torchvision's rule: outside [-1, H] x [-1, W] -> 0, else clamp and interpolate."""
    H, W, _ = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return None
    y, x = max(y, 0.0), max(x, 0.0)
    y0, x0 = int(y), int(x)
    if y0 >= H - 1:
        y1 = y0 = H - 1
        y = float(y0)
    else:
        y1 = y0 + 1
    if x0 >= W - 1:
        x1 = x0 = W - 1
        x = float(x0)
    else:
        x1 = x0 + 1
    ly, lx = y - y0, x - x0
    hy, hx = 1.0 - ly, 1.0 - lx
    return (y0, x0, y1, x1), (hy * hx, hy * lx, ly * hx, ly * lx)


def _geom(roi, scale, PH, PW, sr, aligned):
    off = 0.5 if aligned else 0.0
    x1, y1 = roi[1] * scale - off, roi[2] * scale - off
    rw, rh = roi[3] * scale - off - x1, roi[4] * scale - off - y1
    if not aligned:
        rw, rh = max(rw, 1.0), max(rh, 1.0)
    gh = sr if sr > 0 else int(math.ceil(rh / PH))
    gw = sr if sr > 0 else int(math.ceil(rw / PW))
    return int(roi[0]), y1, x1, rh / PH, rw / PW, gh, gw


def roi_align(x, rois, spatial_scale, output_size, sampling_ratio=2, aligned=False):
    """x (N, H, W, C) float; rois (R, 5) = (batch index, x1, y1, x2, y2) -> (R, PH, PW, C) float64."""
    PH, PW = output_size
    x = np.asarray(x, np.float64)
    out = np.zeros((len(rois), PH, PW, x.shape[-1]))
    for r, roi in enumerate(np.asarray(rois, np.float64)):
        n, y1, x1, bh, bw, gh, gw = _geom(roi, spatial_scale, PH, PW, sampling_ratio, aligned)
        for ph in range(PH):
            for pw in range(PW):
                acc = np.zeros(x.shape[-1])
                for iy in range(gh):
                    yy = y1 + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = x1 + pw * bw + (ix + 0.5) * bw / gw
                        b = _bilinear(x[n], yy, xx)
                        if b is None:
                            continue
                        (y0, x0, y1_, x1_), (w00, w01, w10, w11) = b
                        acc += w00 * x[n, y0, x0] + w01 * x[n, y0, x1_] + w10 * x[n, y1_, x0] + w11 * x[n, y1_, x1_]
                out[r, ph, pw] = acc / max(gh * gw, 1)
    return out


def roi_align_backward(dout, shape, rois, spatial_scale, output_size, sampling_ratio=2, aligned=False):
    """Adjoint of roi_align: dout (R, PH, PW, C) -> dx `shape` = (N, H, W, C)."""
    PH, PW = output_size
    dx = np.zeros(shape)
    dout = np.asarray(dout, np.float64)
    for r, roi in enumerate(np.asarray(rois, np.float64)):
        n, y1, x1, bh, bw, gh, gw = _geom(roi, spatial_scale, PH, PW, sampling_ratio, aligned)
        for ph in range(PH):
            for pw in range(PW):
                g = dout[r, ph, pw] / max(gh * gw, 1)
                for iy in range(gh):
                    yy = y1 + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = x1 + pw * bw + (ix + 0.5) * bw / gw
                        b = _bilinear(dx[n], yy, xx)
                        if b is None:
                            continue
                        (y0, x0, y1_, x1_), (w00, w01, w10, w11) = b
                        dx[n, y0, x0] += w00 * g
                        dx[n, y0, x1_] += w01 * g
                        dx[n, y1_, x0] += w10 * g
                        dx[n, y1_, x1_] += w11 * g
    return dx


def fpn_merge(lateral, top):
    """lateral (N, H, W, C) + nearest-neighbour 2x upsampling of top (N, ceil(H/2), ceil(W/2), C)."""
    lateral, top = np.asarray(lateral, np.float64), np.asarray(top, np.float64)
    H, W = lateral.shape[1:3]
    up = np.repeat(np.repeat(top, 2, axis=1), 2, axis=2)[:, :H, :W]
    return lateral + up


def fpn_merge_backward_top(dout):
    """Gradient w.r.t. `top`: every coarse pixel sums its (up to) 2x2 children."""
    dout = np.asarray(dout, np.float64)
    N, H, W, C = dout.shape
    Ht, Wt = (H + 1) // 2, (W + 1) // 2
    pad = np.zeros((N, 2 * Ht, 2 * Wt, C))
    pad[:, :H, :W] = dout
    return pad.reshape(N, Ht, 2, Wt, 2, C).sum(axis=(2, 4))


# ---------------------------------------------------------------- region proposals (Faster R-CNN conventions)
def decode_boxes(anchors, deltas, image_size=None):
    """float64: boxes = decode(anchors (A, 4) repeating, deltas (k A, 4)), weights 1, dw / dh clamped at log(1000/16)."""
    anchors = np.asarray(anchors, np.float64).reshape(-1, 4)
    deltas = np.asarray(deltas, np.float64).reshape(-1, 4)
    a = np.tile(anchors, (len(deltas) // len(anchors), 1))
    w, h = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    cx, cy = a[:, 0] + 0.5 * w, a[:, 1] + 0.5 * h
    dw, dh = np.minimum(deltas[:, 2], np.log(1000.0 / 16)), np.minimum(deltas[:, 3], np.log(1000.0 / 16))
    pcx, pcy, pw, ph = deltas[:, 0] * w + cx, deltas[:, 1] * h + cy, np.exp(dw) * w, np.exp(dh) * h
    b = np.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], 1)
    if image_size is not None:
        b[:, 0::2] = np.clip(b[:, 0::2], 0, image_size[1])
        b[:, 1::2] = np.clip(b[:, 1::2], 0, image_size[0])
    return b


def box_iou(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes, scores, thr):
    """Greedy NMS: visit boxes in descending (stable) score order, keep a box unless a kept one overlaps it by IoU > thr."""
    boxes = np.asarray(boxes, np.float32).astype(np.float64)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if len(order) == 0:
        return np.zeros(0, np.int64)
    iou = box_iou(boxes[order], boxes[order])
    alive = np.ones(len(order), bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(order[i])
        alive[i + 1:] &= ~(iou[i, i + 1:] > thr)
    return np.array(keep, np.int64)


def rpn_loss(head, labels, targets, A, beta=1.0 / 9):
    """float64 loss terms and gradient of their sum w.r.t. head (P, 5 A); labels (P A,) in {1, 0, -1}; targets (P A, 4)."""
    head = np.asarray(head, np.float64).reshape(-1, 5 * A)
    P = len(head)
    lab = np.asarray(labels).reshape(P, A)
    tgt = np.asarray(targets, np.float64).reshape(P, A, 4)
    x = head[:, :A]
    d = head[:, A:].reshape(P, A, 4)
    n = max(int((lab >= 0).sum()), 1)
    samp, pos = lab >= 0, lab > 0
    t = pos.astype(np.float64)
    bce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    l_obj = (bce * samp).sum() / n
    e = (d - tgt) * pos[..., None]
    ae = np.abs(e)
    sl1 = np.where(ae < beta, 0.5 * e * e / beta, ae - 0.5 * beta)
    l_box = sl1.sum() / n
    g = np.zeros_like(head)
    g[:, :A] = (1.0 / (1.0 + np.exp(-x)) - t) * samp / n
    g[:, A:] = (np.where(ae < beta, e / beta, np.sign(e)) / n).reshape(P, 4 * A)
    return l_obj, l_box, g


def anchor_match(anchors, gt, hi=0.7, lo=0.3, allow_low_quality=True):
    """Matcher + BoxCoder.encode of the usual RPN implementation, float32 IoUs (the comparison with the thresholds and the
    equality of the low-quality rule are decided on the values the device computes: same formula, same order)."""
    a = np.asarray(anchors, np.float32).reshape(-1, 4)
    g = np.asarray(gt, np.float32).reshape(-1, 4)
    n = len(a)
    if len(g) == 0:
        return np.zeros(n, np.int8), np.full(n, -1, np.int32), np.zeros((n, 4), np.float32)
    f = np.float32
    iw = np.maximum(np.minimum(g[:, None, 2], a[None, :, 2]) - np.maximum(g[:, None, 0], a[None, :, 0]), f(0))
    ih = np.maximum(np.minimum(g[:, None, 3], a[None, :, 3]) - np.maximum(g[:, None, 1], a[None, :, 1]), f(0))
    inter = (iw * ih).astype(f)
    area_g = ((g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])).astype(f)
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).astype(f)
    iou = (inter / ((area_g[:, None] + area_a[None, :]).astype(f) - inter).astype(f)).astype(f)        # (g, n)
    mv, mi = iou.max(0), iou.argmax(0)
    lab = np.where(mv >= f(hi), 1, np.where(mv < f(lo), 0, -1)).astype(np.int8)
    if allow_low_quality:
        best = iou.max(1)
        lq = ((iou == best[:, None]) & (iou > 0)).any(0)
        lab[lq] = 1
    matched = np.where(lab == 1, mi, -1).astype(np.int32)
    t = np.zeros((n, 4), np.float64)
    pos = lab == 1
    ga, aa = g[matched[pos]].astype(np.float64), a[pos].astype(np.float64)
    aw, ah, gw, gh = aa[:, 2] - aa[:, 0], aa[:, 3] - aa[:, 1], ga[:, 2] - ga[:, 0], ga[:, 3] - ga[:, 1]
    t[pos] = np.stack([((ga[:, 0] + 0.5 * gw) - (aa[:, 0] + 0.5 * aw)) / aw, ((ga[:, 1] + 0.5 * gh) - (aa[:, 1] + 0.5 * ah)) / ah,
                       np.log(gw / aw), np.log(gh / ah)], 1)
    return lab, matched, t
