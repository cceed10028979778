#!/usr/bin/env python3
"""One training step of the Mask R-CNN PIECES built so far, chained on the GPU (BASELINE configs[3] shape:
batch 64 x 128 x 128 x 3 patches): ResNet-50-FPN backbone -> RPN head on P3 (stride 8) with its loss -> RoIAlign of the
ground-truth boxes on P2 (stride 4): 7 x 7 -> box head with the Fast R-CNN loss, 14 x 14 -> mask head with its loss ->
gradients back through RoIAlign and the RPN head into the pyramid -> backbone backward -> clip + Adam of the four
parameter sets.  NOT a full detector: the RoIs are the ground-truth boxes (no proposal sampling), the RPN runs on one
pyramid level, anchor labels / regression targets of the (static, synthetic) boxes are prepared once.  Every
tensor stays in HBM; one host sync per step (the two RPN loss scalars).

    python tools/bench_maskrcnn_lite.py [--batch 64] [--rois-per-image 4] [--dtype f32|bf16]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rois-per-image", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    a = ap.parse_args()
    import torch
    from rfi_toolbox_amd._lib import DEVICE, Hyper, check, lib
    from rfi_toolbox_amd.models import BoxHead, MaskHead, ResNet50FPN, RPNHead
    from rfi_toolbox_amd.models import detection_ops as ops
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    torch.manual_seed(0)
    mode = "float32" if a.dtype == "f32" else "bfloat16"
    n, s, F, A, k = a.batch, 128, 256, 4, a.rois_per_image
    backbone = ResNet50FPN(3, 64, F).set_compute_dtype(mode)
    rpn = RPNHead(F, A, 1).train().set_compute_dtype(mode)
    mask = MaskHead(F, 1, 4).train().set_compute_dtype(mode)
    box = BoxHead(F, 7, 1024, 2).train().set_compute_dtype(mode)
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((n, s, s, 3)).astype(np.float32))
    # synthetic ground truth: k boxes per image, their masks; RPN anchors of 4 sizes on the stride-8 grid
    x1 = rng.uniform(0, 80, (n, k)); y1 = rng.uniform(0, 80, (n, k))
    gt = np.stack([x1, y1, x1 + rng.uniform(12, 48, (n, k)), y1 + rng.uniform(12, 48, (n, k))], -1).astype(np.float32)
    rois = np.concatenate([np.repeat(np.arange(n), k)[:, None].astype(np.float32), gt.reshape(-1, 4)], 1)
    R = len(rois)
    d_rois = ctx.to_device(rois)
    d_mask_t = ctx.to_device((rng.random((R, 28, 28)) > 0.5).astype(np.uint8))
    g3 = s // 8
    sizes = np.array([16, 32, 64, 128], np.float32)
    ys, xs = np.meshgrid(np.arange(g3) * 8 + 4, np.arange(g3) * 8 + 4, indexing="ij")
    anchors = np.stack([xs[..., None] - sizes / 2, ys[..., None] - sizes / 2, xs[..., None] + sizes / 2, ys[..., None] + sizes / 2],
                       -1).reshape(-1, 4).astype(np.float32)
    labels, targets = [], []
    for i in range(n):                                  # (once: the boxes are static)
        lab, _, tgt = ops.anchor_match(anchors, gt[i])
        neg = np.flatnonzero(lab == 0)
        lab[neg[rng.permutation(len(neg))[128:]]] = -1  # sample at most 128 negatives per image
        labels.append(lab); targets.append(tgt)
    labels, targets = np.concatenate(labels), np.concatenate(targets)
    d_lab, d_tgt = ctx.to_device(labels), ctx.to_device(targets)
    n_sampled = int((labels >= 0).sum())
    shapes = [(n, s >> (2 + i), s >> (2 + i), F) for i in range(5)]
    feats = [ctx.empty(sh, np.float32) for sh in shapes]
    dfeat2, dfeat3 = ctx.empty(shapes[0], np.float32), ctx.empty(shapes[1], np.float32)
    pf = (C.c_void_p * 5)(*[f.ptr for f in feats])
    pdf = (C.c_void_p * 5)(dfeat2.ptr, dfeat3.ptr, None, None, None)
    rpn_out = ctx.empty((n, g3, g3, 5 * A), np.float32)
    rpn_dout = ctx.empty((n, g3, g3, 5 * A), np.float32)
    roi_feats = ctx.empty((R, 14, 14, F), np.float32)
    roi_grad = ctx.empty((R, 14, 14, F), np.float32)
    roi7, roi7_grad = ctx.empty((R, 7, 7, F), np.float32), ctx.empty((R, 7, 7, F), np.float32)
    box_out, box_dout = ctx.empty((R, 10), np.float32), ctx.empty((R, 10), np.float32)
    d_box_lab = ctx.to_device(np.ones(R, np.int32))                       # every RoI is a ground-truth box of class 1
    d_box_tgt = ctx.to_device(np.zeros((R, 4), np.float32))
    dfeat2b = ctx.empty(shapes[0], np.float32)
    lc, lr = C.c_float(), C.c_float()
    hp = Hyper(1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0)
    lo, lb, lm, nrm = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    P = lambda d: C.c_void_p(d.ptr)

    def step():
        check(lib.rfi_backbone_forward(backbone._h, P(x), DEVICE, n, s, s, pf, DEVICE))
        # RPN on P3
        check(lib.rfi_model_forward_nhwc(rpn._h, P(feats[1]), DEVICE, n, g3, g3, P(rpn_out), DEVICE))
        check(lib.rfi_op_rpn_loss(ctx.handle, P(rpn_out), n * g3 * g3, A, P(d_lab), P(d_tgt), n_sampled, 1.0 / 9, P(rpn_dout),
                                  C.byref(lo), C.byref(lb)))
        check(lib.rfi_model_backward_dlogits(rpn._h, P(feats[1]), DEVICE, P(rpn_dout), DEVICE, n, g3, g3))
        check(lib.rfi_model_input_grad(rpn._h, P(dfeat3), DEVICE))
        # mask branch on P2
        check(lib.rfi_op_roi_align(ctx.handle, P(feats[0]), n, s // 4, s // 4, F, P(d_rois), R, 0.25, 14, 14, 2, 0, P(roi_feats)))
        check(lib.rfi_train_forward_backward(mask._h, P(roi_feats), DEVICE, P(d_mask_t), DEVICE, R, 14, 14, C.byref(lm)))
        check(lib.rfi_model_input_grad(mask._h, P(roi_grad), DEVICE))
        check(lib.rfi_op_roi_align_backward(ctx.handle, P(roi_grad), n, s // 4, s // 4, F, P(d_rois), R, 0.25, 14, 14, 2, 0, P(dfeat2)))
        # box branch on P2
        check(lib.rfi_op_roi_align(ctx.handle, P(feats[0]), n, s // 4, s // 4, F, P(d_rois), R, 0.25, 7, 7, 2, 0, P(roi7)))
        check(lib.rfi_model_forward_nhwc(box._h, P(roi7), DEVICE, R, 1, 1, P(box_out), DEVICE))
        check(lib.rfi_op_fastrcnn_loss(ctx.handle, P(box_out), R, 2, P(d_box_lab), P(d_box_tgt), 1.0 / 9, P(box_dout), C.byref(lc),
                                       C.byref(lr)))
        check(lib.rfi_model_backward_dlogits(box._h, P(roi7), DEVICE, P(box_dout), DEVICE, R, 1, 1))
        check(lib.rfi_model_input_grad(box._h, P(roi7_grad), DEVICE))
        check(lib.rfi_op_roi_align_backward(ctx.handle, P(roi7_grad), n, s // 4, s // 4, F, P(d_rois), R, 0.25, 7, 7, 2, 0, P(dfeat2b)))
        check(lib.rfi_op_add_inplace(ctx.handle, P(dfeat2), P(dfeat2b), n * (s // 4) * (s // 4) * F))
        check(lib.rfi_backbone_backward(backbone._h, P(x), DEVICE, n, s, s, pdf, DEVICE))
        for m in (backbone, rpn, mask, box):
            check(lib.rfi_train_apply(m._h, C.byref(hp), 1.0, C.byref(nrm)))
        return lo.value + lb.value + lm.value + lc.value + lr.value

    losses = [step() for _ in range(3)]
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(step())
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "Mask R-CNN pieces, training patches/s (backbone + RPN on P3 + RoIAlign + box head + mask head)",
                      "value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 2), "batch": n, "rois": R, "dtype": a.dtype,
                      "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4)}))


if __name__ == "__main__":
    main()
