"""Segmentation metrics with the reference's signatures and edge-case rules
(rfi_toolbox/evaluation/metrics.py:25-172).  The three confusion counts (TP, FP, FN) are ONE
reduction on the GPU (uint8 or float32 input, non-zero == positive, metrics.py:36-37); the
ratios are formed on the host in Python floats exactly as the reference forms them."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import DEVICE, FLOAT32, HOST, U8, check, lib
from ..runtime import Context, DeviceArray, is_torch


def _operand(a, ctx):
    """-> (ptr, dtype_code, mem, count, keepalive)"""
    if isinstance(a, DeviceArray):
        code = {np.dtype(np.uint8): U8, np.dtype(np.float32): FLOAT32}[a.dtype]
        return a.ptr, code, DEVICE, int(np.prod(a.shape, dtype=np.int64)), a
    if is_torch(a):
        import torch
        t = a.detach()
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        if t.dtype not in (torch.uint8, torch.float32):
            t = (t != 0).to(torch.uint8)
        t = t.contiguous()
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
            return t.data_ptr(), (U8 if t.dtype == torch.uint8 else FLOAT32), DEVICE, t.numel(), t
        a = t.numpy()
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8) if a.flags.c_contiguous else a.astype(np.uint8)
    elif a.dtype not in (np.uint8, np.float32):
        a = (a != 0).astype(np.uint8)
    a = np.ascontiguousarray(a)
    return a.ctypes.data, (U8 if a.dtype == np.uint8 else FLOAT32), HOST, a.size, a


def confusion_counts(pred, true, device=None):
    """(tp, fp, fn) as Python ints."""
    ctx = Context.get(device)
    pp, pd, pm, pn, k1 = _operand(pred, ctx)
    tp_, td, tm, tn, k2 = _operand(true, ctx)
    if pn != tn:
        raise ValueError(f"pred has {pn} elements, true has {tn}")
    tp, fp, fn = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.rfi_confusion_counts(ctx.handle, C.c_void_p(pp), pd, pm, C.c_void_p(tp_), td, tm, pn,
                                   C.byref(tp), C.byref(fp), C.byref(fn)))
    del k1, k2
    return tp.value, fp.value, fn.value


def _iou(tp, fp, fn):
    union = tp + fp + fn
    return 1.0 if union == 0 else tp / union


def _precision(tp, fp, fn):
    if tp + fp == 0:
        return 1.0 if fn == 0 else 0.0
    return tp / (tp + fp)


def _recall(tp, fp, fn):
    return 1.0 if tp + fn == 0 else tp / (tp + fn)


def _f1(tp, fp, fn):
    p, r = _precision(tp, fp, fn), _recall(tp, fp, fn)
    return 0.0 if p + r == 0 else 2 * (p * r) / (p + r)


def _dice(tp, fp, fn):
    return 1.0 if 2 * tp + fp + fn == 0 else (2 * tp) / (2 * tp + fp + fn)


def compute_iou(pred, true):
    return _iou(*confusion_counts(pred, true))


def compute_precision(pred, true):
    return _precision(*confusion_counts(pred, true))


def compute_recall(pred, true):
    return _recall(*confusion_counts(pred, true))


def compute_f1(pred, true):
    return _f1(*confusion_counts(pred, true))


def compute_dice(pred, true):
    return _dice(*confusion_counts(pred, true))


def evaluate_segmentation(pred, true):
    """dict with 'iou', 'precision', 'recall', 'f1', 'dice' (metrics.py:155-172)."""
    c = confusion_counts(pred, true)
    return {"iou": _iou(*c), "precision": _precision(*c), "recall": _recall(*c), "f1": _f1(*c),
            "dice": _dice(*c)}
