#!/usr/bin/env python3
"""Overfit the detector assembly (``MaskRCNN``) on a few synthetic patches with rectangular bright regions and report how
well ``predict`` recovers them: best box IoU per ground-truth region and the IoU of the union mask.  A functional
demonstration (the pieces are parity-tested one by one), not a benchmark.

    python tools/demo_mask_rcnn.py [--steps 200] [--dtype float32|bfloat16]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def batch(rng, n, size):
    x = rng.standard_normal((n, size, size, 3)).astype(np.float32) * 0.1
    targets = []
    for i in range(n):
        boxes, masks = [], []
        for _ in range(2):
            w, h = rng.integers(20, 60, 2)
            x1, y1 = rng.integers(0, size - w), rng.integers(0, size - h)
            m = np.zeros((size, size), np.uint8)
            m[y1:y1 + h, x1:x1 + w] = 1
            x[i, y1:y1 + h, x1:x1 + w] += 2.0
            boxes.append([x1, y1, x1 + w, y1 + h]); masks.append(m)
        targets.append({"boxes": np.asarray(boxes, np.float32), "labels": np.ones(2, np.int64), "masks": np.stack(masks)})
    return x, targets


def iou(a, b):
    ix = max(0.0, min(a[2], b[2]) - max(a[0], b[0])) * max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    return ix / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - ix)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--images", type=int, default=4)
    a = ap.parse_args()
    import torch
    from rfi_toolbox_amd.models import MaskRCNN
    torch.manual_seed(0)
    det = MaskRCNN(2, 3, 16, 64, 128, seed=0).set_compute_dtype(a.dtype)
    x, targets = batch(np.random.default_rng(1), a.images, 128)
    t0 = time.time()
    for s in range(a.steps):
        l = det.train_step(x, targets, lr=2e-3, weight_decay=0.0, max_grad_norm=10.0)
        if s % 20 == 0 or s == a.steps - 1:
            print(f"step {s:4d}  " + "  ".join(f"{k[5:] or 'total'} {v:.4f}" for k, v in l.items()), flush=True)
    print(f"{a.steps} steps in {time.time() - t0:.1f} s")
    det.score_thresh = 0.5
    out = det.predict(x)
    for i, (o, t) in enumerate(zip(out, targets)):
        best = [max((iou(g, b) for b in o["boxes"]), default=0.0) for g in t["boxes"]]
        gm = t["masks"].any(0)
        mi = (gm & o["rfi_mask"]).sum() / max((gm | o["rfi_mask"]).sum(), 1)
        print(f"image {i}: {len(o['boxes'])} detections, best box IoU per region {[round(float(b), 3) for b in best]}, union-mask IoU {mi:.3f}")


if __name__ == "__main__":
    main()
