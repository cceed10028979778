"""NumPy oracle for ``evaluate_segmentation``.  TEST INFRASTRUCTURE ONLY.

Restates rfi_toolbox/evaluation/metrics.py:25-172 from one confusion count
(tp, fp, fn) instead of five independent passes; edge-case returns follow the
reference line by line:

* IoU        1.0 when the union is empty                    metrics.py:42-43
* precision  no predictions: 1.0 if fn == 0 else 0.0        metrics.py:70-77
* recall     1.0 when there are no positives                metrics.py:101-102
* F1         0.0 when precision + recall == 0               metrics.py:123-124
* dice       1.0 when 2tp+fp+fn == 0                        metrics.py:149-150

Inputs are "non-zero is True" (metrics.py:36-37).
"""
from __future__ import annotations

import numpy as np


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def confusion(pred, true):
    p = _np(pred).astype(bool)
    t = _np(true).astype(bool)
    tp = int(np.count_nonzero(p & t))
    fp = int(np.count_nonzero(p & ~t))
    fn = int(np.count_nonzero(~p & t))
    return tp, fp, fn


def metrics_from_counts(tp, fp, fn):
    union = tp + fp + fn
    iou = 1.0 if union == 0 else tp / union
    if tp + fp == 0:
        precision = 1.0 if fn == 0 else 0.0
    else:
        precision = tp / (tp + fp)
    recall = 1.0 if tp + fn == 0 else tp / (tp + fn)
    f1 = 0.0 if precision + recall == 0 else 2 * (precision * recall) / (precision + recall)
    dice = 1.0 if 2 * tp + fp + fn == 0 else (2 * tp) / (2 * tp + fp + fn)
    return {"iou": iou, "precision": precision, "recall": recall, "f1": f1, "dice": dice}


def evaluate_segmentation(pred, true):
    return metrics_from_counts(*confusion(pred, true))
