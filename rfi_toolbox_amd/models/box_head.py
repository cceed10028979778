"""``BoxHead`` -- the box branch of Faster / Mask R-CNN (BASELINE.json configs[3]).

Not in the reference (it contains no detector) and torchvision is absent: builder-defined as the published head (layer
names of torchvision's ``TwoMLPHead`` + ``FastRCNNPredictor``):

    fc6 = Linear(C * 7 * 7, 1024) -> ReLU;  fc7 = Linear(1024, 1024) -> ReLU
    cls_score = Linear(1024, K1);  bbox_pred = Linear(1024, 4 K1)            # K1 = classes incl. background

on RoIAlign-ed features.  The library takes the RoI features as they come out of ``detection_ops.roi_align`` -- NHWC,
``(R, 7, 7, C)`` -- and flattens them in (h, w, c) order; this class permutes ``fc6.weight`` from torch's (c, h, w) column
order at the ``state_dict`` boundary and stacks ``cls_score`` / ``bbox_pred`` into the library's one output layer.

    out = head.train().forward_rois(roi_feats)                   # (R, 5 K1): K1 class logits, then K1 x 4 deltas
    l_cls, l_box, dout = detection_ops.fastrcnn_loss(out, labels, targets)
    head.backward(roi_feats, dout); head.apply_gradients(...)    # input_grad() feeds roi_align_backward
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from .._lib import HOST, check, lib
from ..runtime import as_pointer, torch
from .unet import HipSegmenter, default_init_state


class BoxHead(HipSegmenter):
    _first_key = "fc6.weight"

    def __init__(self, in_channels=256, resolution=7, representation_size=1024, num_classes=2, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (resolution, "resolution"), (representation_size, "representation_size"),
                      (num_classes, "num_classes")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if in_channels % 4 or representation_size % 4 or num_classes < 2:
            raise ValueError("in_channels and representation_size must be multiples of 4, num_classes >= 2 (background + 1)")
        self.in_channels, self.resolution, self.hidden, self.num_classes = int(in_channels), int(resolution), int(representation_size), int(num_classes)
        self.in_features = self.in_channels * self.resolution ** 2
        self.out_channels = 5 * self.num_classes
        d, h, k = self.in_features, self.hidden, self.num_classes
        self._entries = [("fc6.weight", (h, d, 1, 1), "conv_w"), ("fc6.bias", (h,), "conv_b"), ("fc7.weight", (h, h, 1, 1), "conv_w"),
                         ("fc7.bias", (h,), "conv_b"), ("head.weight", (5 * k, h, 1, 1), "conv_w"), ("head.bias", (5 * k,), "conv_b")]
        # torch.nn construction order and shapes (Linear: kaiming_uniform(a = sqrt 5) on (out, in), as Conv2d on (out, in, 1, 1))
        split = [("fc6.weight", (h, d, 1, 1), "conv_w"), ("fc6.bias", (h,), "conv_b"), ("fc7.weight", (h, h, 1, 1), "conv_w"),
                 ("fc7.bias", (h,), "conv_b"), ("cls_score.weight", (k, h, 1, 1), "conv_w"), ("cls_score.bias", (k,), "conv_b"),
                 ("bbox_pred.weight", (4 * k, h, 1, 1), "conv_w"), ("bbox_pred.bias", (4 * k,), "conv_b")]
        init = default_init_state(0, 0, 0, entries=split)
        self._init = self._to_lib(OrderedDict((n, v.reshape(v.shape[:2]) if v.ndim == 4 else v) for n, v in init.items()))
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_box_head_create(ctx.handle, self.in_features, self.hidden, 2, self.out_channels, C.byref(h)))
        return h

    # ---- torch layout (Linear weights, (c, h, w) feature order, two predictors) <-> library layout
    def _to_lib(self, sd):
        if "head.weight" in sd:
            return sd
        t = torch.as_tensor
        c, r = self.in_channels, self.resolution
        out = OrderedDict()
        out["fc6.weight"] = t(sd["fc6.weight"]).reshape(self.hidden, c, r, r).permute(0, 2, 3, 1).reshape(self.hidden, -1, 1, 1).contiguous()
        out["fc6.bias"] = t(sd["fc6.bias"])
        out["fc7.weight"] = t(sd["fc7.weight"]).reshape(self.hidden, self.hidden, 1, 1)
        out["fc7.bias"] = t(sd["fc7.bias"])
        out["head.weight"] = torch.cat([t(sd["cls_score.weight"]), t(sd["bbox_pred.weight"])], 0).reshape(self.out_channels, self.hidden, 1, 1)
        out["head.bias"] = torch.cat([t(sd["cls_score.bias"]), t(sd["bbox_pred.bias"])], 0)
        return out

    def _from_lib(self, name, v):
        c, r, k = self.in_channels, self.resolution, self.num_classes
        if name == "fc6.weight":
            return v.reshape(self.hidden, r, r, c).permute(0, 3, 1, 2).reshape(self.hidden, -1)
        return v.reshape(v.shape[0], -1) if v.ndim == 4 else v

    def state_dict(self):
        sd = super().state_dict()
        k = self.num_classes
        out = OrderedDict()
        for n in ("fc6.weight", "fc6.bias", "fc7.weight", "fc7.bias"):
            out[n] = self._from_lib(n, sd[n]).clone()
        hw, hb = self._from_lib("head.weight", sd["head.weight"]), sd["head.bias"]
        out["cls_score.weight"], out["cls_score.bias"] = hw[:k].clone(), hb[:k].clone()
        out["bbox_pred.weight"], out["bbox_pred.bias"] = hw[k:].clone(), hb[k:].clone()
        return out

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict(self._to_lib(OrderedDict(state_dict)), strict)

    def grad(self, name):
        k = self.num_classes
        t = torch.from_numpy
        for pre, sl in (("cls_score.", slice(0, k)), ("bbox_pred.", slice(k, 5 * k))):
            if name.startswith(pre):
                return self._from_lib("head." + name[len(pre):], t(super().grad("head." + name[len(pre):]))).numpy()[sl]
        return self._from_lib(name, t(super().grad(name))).numpy()

    # ---- forward / backward on RoI features (R, res, res, C) NHWC
    def forward_rois(self, roi_feats):
        r = roi_feats.shape[0]
        x = np.ascontiguousarray(np.asarray(roi_feats, np.float32).reshape(r, 1, 1, self.in_features))
        xp, xm, keep = as_pointer(x, np.float32, self.ctx)
        out = np.empty((r, self.out_channels), np.float32)
        check(lib.rfi_model_forward_nhwc(self._h, C.c_void_p(xp), xm, r, 1, 1, out.ctypes.data_as(C.c_void_p), HOST))
        del keep
        return out

    def backward(self, roi_feats, dout):
        r = roi_feats.shape[0]
        x = np.ascontiguousarray(np.asarray(roi_feats, np.float32).reshape(r, 1, 1, self.in_features))
        xp, xm, keep = as_pointer(x, np.float32, self.ctx)
        d = np.ascontiguousarray(np.asarray(dout, np.float32).reshape(r, self.out_channels))
        check(lib.rfi_model_backward_dlogits(self._h, C.c_void_p(xp), xm, d.ctypes.data_as(C.c_void_p), HOST, r, 1, 1))
        del keep

    def input_grad(self, shape) -> np.ndarray:
        out = np.empty(tuple(shape), dtype=np.float32)
        check(lib.rfi_model_input_grad(self._h, out.ctypes.data_as(C.c_void_p), HOST))
        return out
