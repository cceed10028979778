"""Building blocks of the Mask R-CNN path named by BASELINE.json configs[3]: RoIAlign and the FPN
top-down merge as HIP kernels behind the C ABI (include/rfi_hip.h).  Not in the reference (it has no detector code):
builder-defined from the published algorithms; NHWC float32 feature maps, device-resident (`DeviceArray`) or NumPy."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import check, lib
from ..runtime import Context, DeviceArray


def _dev(ctx, a):
    return a if isinstance(a, DeviceArray) else ctx.to_device(np.ascontiguousarray(a, np.float32))


def _p(d):
    return C.c_void_p(d.ptr)


def roi_align(x, rois, spatial_scale=1.0, output_size=(7, 7), sampling_ratio=2, aligned=False, device=None, to_host=True):
    """x (N, H, W, C) float32, C % 4 == 0; rois (R, 5) = (batch index, x1, y1, x2, y2) -> (R, PH, PW, C)."""
    ctx = Context.get(device)
    dx, dr = _dev(ctx, x), _dev(ctx, np.asarray(rois, np.float32).reshape(-1, 5))
    n, h, w, c = dx.shape
    ph, pw = output_size
    out = ctx.empty((dr.shape[0], ph, pw, c), np.float32)
    check(lib.rfi_op_roi_align(ctx.handle, _p(dx), n, h, w, c, _p(dr), dr.shape[0], float(spatial_scale), ph, pw,
                               int(sampling_ratio), 1 if aligned else 0, _p(out)))
    ctx.synchronize()
    return out.numpy() if to_host else out


def roi_align_backward(dout, input_shape, rois, spatial_scale=1.0, sampling_ratio=2, aligned=False, device=None, to_host=True):
    ctx = Context.get(device)
    dd, dr = _dev(ctx, dout), _dev(ctx, np.asarray(rois, np.float32).reshape(-1, 5))
    n, h, w, c = input_shape
    r, ph, pw, _ = dd.shape
    dx = ctx.empty((n, h, w, c), np.float32)
    check(lib.rfi_op_roi_align_backward(ctx.handle, _p(dd), n, h, w, c, _p(dr), r, float(spatial_scale), ph, pw,
                                        int(sampling_ratio), 1 if aligned else 0, _p(dx)))
    ctx.synchronize()
    return dx.numpy() if to_host else dx


def fpn_merge(lateral, top, device=None, to_host=True):
    """lateral (N, H, W, C) + nearest-neighbour 2x upsampling of top (N, ceil(H/2), ceil(W/2), C)."""
    ctx = Context.get(device)
    dl, dt = _dev(ctx, lateral), _dev(ctx, top)
    n, h, w, c = dl.shape
    if dt.shape != (n, (h + 1) // 2, (w + 1) // 2, c):
        raise ValueError(f"top must be {(n, (h + 1) // 2, (w + 1) // 2, c)}, got {dt.shape}")
    out = ctx.empty((n, h, w, c), np.float32)
    check(lib.rfi_op_fpn_merge(ctx.handle, _p(dl), _p(dt), n, h, w, c, _p(out)))
    ctx.synchronize()
    return out.numpy() if to_host else out


def fpn_merge_backward(dout, device=None, to_host=True):
    """-> (d_lateral, d_top): d_lateral is dout itself, d_top sums each coarse pixel's 2x2 children."""
    ctx = Context.get(device)
    dd = _dev(ctx, dout)
    n, h, w, c = dd.shape
    dtop = ctx.empty((n, (h + 1) // 2, (w + 1) // 2, c), np.float32)
    check(lib.rfi_op_fpn_merge_backward(ctx.handle, _p(dd), n, h, w, c, _p(dtop)))
    ctx.synchronize()
    return (dd.numpy(), dtop.numpy()) if to_host else (dd, dtop)


# ---------------------------------------------------------------- region proposals (rpn_kernels.hip)
def decode_boxes(anchors, deltas, image_size=None, device=None):
    """boxes = decode(anchors (A, 4), deltas (k A, 4)) with the Faster R-CNN parameterisation (weights 1, dw / dh clamped at
    log(1000/16)); clipped to ``image_size = (H, W)`` when given.  -> (k A, 4) float32."""
    ctx = Context.get(device)
    da = _dev(ctx, np.asarray(anchors, np.float32).reshape(-1, 4))
    dd = _dev(ctx, np.asarray(deltas, np.float32).reshape(-1, 4))
    n, na = dd.shape[0], da.shape[0]
    if n % na:
        raise ValueError(f"{n} delta rows are not a multiple of {na} anchors")
    out = ctx.empty((n, 4), np.float32)
    h, w = (float(image_size[0]), float(image_size[1])) if image_size is not None else (0.0, 0.0)
    check(lib.rfi_op_box_decode(ctx.handle, _p(da), na, _p(dd), n, h, w, _p(out)))
    ctx.synchronize()
    return out.numpy()


def nms(boxes, scores, iou_threshold, device=None):
    """Greedy non-maximum suppression: indices of the kept boxes, in descending score order (stable for ties)."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    scores = np.asarray(scores, np.float32).reshape(-1)
    if len(boxes) != len(scores):
        raise ValueError("boxes and scores differ in length")
    if len(boxes) == 0:
        return np.zeros(0, np.int64)
    order = np.argsort(-scores, kind="stable")
    ctx = Context.get(device)
    db = ctx.to_device(np.ascontiguousarray(boxes[order]))
    keep = np.empty(len(boxes), np.int32)
    nk = C.c_int()
    check(lib.rfi_op_nms(ctx.handle, _p(db), len(boxes), float(iou_threshold), keep.ctypes.data_as(C.c_void_p), C.byref(nk)))
    return order[keep[:nk.value]]


def rpn_loss(head, labels, targets, anchors_per_pixel, beta=1.0 / 9, device=None, num_sampled=None):
    """head (P, 5 A) float32 = A objectness logits then A x 4 box deltas per pixel; labels (P A,) int8 in {1, 0, -1};
    targets (P A, 4).  -> (objectness loss, box loss, d(sum)/d(head) (P, 5 A)).  ``num_sampled``: the normaliser (default:
    the sampled anchors of THESE labels; pass the total over all pyramid levels when the loss is evaluated level by level)."""
    ctx = Context.get(device)
    a = int(anchors_per_pixel)
    dh = _dev(ctx, np.asarray(head, np.float32).reshape(-1, 5 * a))
    lab = np.ascontiguousarray(np.asarray(labels, np.int8).reshape(-1))
    if lab.size != dh.shape[0] * a:
        raise ValueError("labels do not match the head output")
    dl = ctx.to_device(lab)
    dt = _dev(ctx, np.asarray(targets, np.float32).reshape(-1, 4))
    dg = ctx.empty(dh.shape, np.float32)
    lo, lb = C.c_float(), C.c_float()
    check(lib.rfi_op_rpn_loss(ctx.handle, _p(dh), dh.shape[0], a, _p(dl), _p(dt),
                              int((lab >= 0).sum()) if num_sampled is None else int(num_sampled), float(beta), _p(dg),
                              C.byref(lo), C.byref(lb)))
    return lo.value, lb.value, dg.numpy()


def anchor_match(anchors, gt_boxes, fg_iou=0.7, bg_iou=0.3, allow_low_quality=True, device=None):
    """RPN training targets of one image: anchors (n, 4) vs ground truth (g, 4) -> labels int8 (n,) in {1, 0, -1}, matched
    ground-truth index int32 (n,) (-1 unless positive), regression targets float32 (n, 4) (zeros unless positive)."""
    ctx = Context.get(device)
    da = _dev(ctx, np.asarray(anchors, np.float32).reshape(-1, 4))
    g = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
    dg = ctx.to_device(np.ascontiguousarray(g if len(g) else np.zeros((1, 4), np.float32)))
    n = da.shape[0]
    labels, matched, targets = ctx.empty((n,), np.int8), ctx.empty((n,), np.int32), ctx.empty((n, 4), np.float32)
    check(lib.rfi_op_anchor_match(ctx.handle, _p(da), n, _p(dg), len(g), float(fg_iou), float(bg_iou), 1 if allow_low_quality else 0,
                                  _p(labels), _p(matched), _p(targets)))
    ctx.synchronize()
    return labels.numpy(), matched.numpy(), targets.numpy()


def fastrcnn_loss(head, labels, targets, beta=1.0 / 9, device=None):
    """head (R, 5 K1) float32 = K1 class logits then K1 x 4 box deltas per RoI; labels (R,) in [0, K1) (0 = background);
    targets (R, 4).  -> (classification loss, box loss, d(sum)/d(head) (R, 5 K1))."""
    ctx = Context.get(device)
    h = np.asarray(head, np.float32)
    r, k1 = h.shape[0], h.shape[1] // 5
    dh = _dev(ctx, h)
    lab = np.ascontiguousarray(np.asarray(labels, np.int32).reshape(-1))
    if lab.size != r or lab.min() < 0 or lab.max() >= k1:
        raise ValueError("labels must be one class index in [0, K1) per RoI")
    dl, dt = ctx.to_device(lab), _dev(ctx, np.asarray(targets, np.float32).reshape(r, 4))
    dg = ctx.empty((r, 5 * k1), np.float32)
    lc, lb = C.c_float(), C.c_float()
    check(lib.rfi_op_fastrcnn_loss(ctx.handle, _p(dh), r, k1, _p(dl), _p(dt), float(beta), _p(dg), C.byref(lc), C.byref(lb)))
    return lc.value, lb.value, dg.numpy()


# ---------------------------------------------------------------- batched forms: one launch for a whole batch of images
def _gt_pack(gt_list):
    """list of (g_i, 4) arrays -> padded (B, Gmax, 4) float32 + counts (B,) int32."""
    gs = [np.asarray(g, np.float32).reshape(-1, 4) for g in gt_list]
    gmax = max(1, max(len(g) for g in gs))
    out = np.zeros((len(gs), gmax, 4), np.float32)
    for i, g in enumerate(gs):
        out[i, :len(g)] = g
    return out, np.asarray([len(g) for g in gs], np.int32)


def anchor_match_batched(anchors, gt_list, fg_iou=0.7, bg_iou=0.3, allow_low_quality=True, anchor_counts=None, device=None):
    """``anchor_match`` for every image of a batch in one launch.  ``anchors``: (n, 4) shared by the images, or (B, n, 4) per
    image with ``anchor_counts`` (B,) valid rows each (rows beyond get label -2).  -> labels int8 (B, n), matched int32
    (B, n), targets float32 (B, n, 4)."""
    ctx = Context.get(device)
    a = np.ascontiguousarray(np.asarray(anchors, np.float32))
    gt, gc = _gt_pack(gt_list)
    B = len(gc)
    shared = a.ndim == 2
    n = a.shape[-2]
    if not shared and a.shape[0] != B:
        raise ValueError("per-image anchors need one block per image")
    da, dg, dgc = ctx.to_device(a), ctx.to_device(gt), ctx.to_device(gc)
    dac = None if anchor_counts is None else ctx.to_device(np.ascontiguousarray(np.asarray(anchor_counts, np.int32)))
    labels, matched, targets = ctx.empty((B, n), np.int8), ctx.empty((B, n), np.int32), ctx.empty((B, n, 4), np.float32)
    check(lib.rfi_op_anchor_match_batched(ctx.handle, _p(da), n, 0 if shared else n, None if dac is None else _p(dac), _p(dg), B,
                                          gt.shape[1], _p(dgc), float(fg_iou), float(bg_iou), 1 if allow_low_quality else 0,
                                          _p(labels), _p(matched), _p(targets)))
    return labels.numpy(), matched.numpy(), targets.numpy()


def nms_batched(boxes_sorted, counts, iou_threshold, device=None):
    """Greedy NMS of B independent sets: ``boxes_sorted`` (B, K, 4) with K <= 256, every set in descending score order and
    ``counts[b]`` valid rows.  -> keep bool (B, K)."""
    ctx = Context.get(device)
    b = np.ascontiguousarray(np.asarray(boxes_sorted, np.float32))
    if b.ndim != 3 or b.shape[2] != 4 or b.shape[1] > 256:
        raise ValueError("boxes_sorted must be (B, K <= 256, 4)")
    c = np.ascontiguousarray(np.asarray(counts, np.int32).reshape(-1))
    if len(c) != b.shape[0]:
        raise ValueError("one count per set")
    if b.shape[0] == 0 or b.shape[1] == 0:
        return np.zeros(b.shape[:2], bool)
    db, dc = ctx.to_device(b), ctx.to_device(c)
    keep = ctx.empty(b.shape[:2], np.uint8)
    check(lib.rfi_op_nms_batched(ctx.handle, _p(db), _p(dc), b.shape[0], b.shape[1], float(iou_threshold), _p(keep)))
    ctx.synchronize()
    return keep.numpy().astype(bool)
