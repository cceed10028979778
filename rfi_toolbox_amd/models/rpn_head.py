"""``RPNHead`` -- the region-proposal head of Faster / Mask R-CNN (BASELINE.json configs[3]).

Not in the reference (it contains no detector) and torchvision is absent: builder-defined as the published head (Ren et
al. 2015; layer names of torchvision's ``RPNHead``):

    conv        = Sequential(Conv2d(C, C, 3, padding=1), ReLU())      # ``conv.0.0.weight`` / ``.bias``
    cls_logits  = Conv2d(C, A, 1)                                       # objectness per anchor
    bbox_pred   = Conv2d(C, 4 A, 1)                                     # deltas, anchor-major (a, 4)

The library runs ``cls_logits`` and ``bbox_pred`` as ONE 1x1 conv with 5 A outputs (``head.weight`` rows [0, A) and
[A, 5 A)); this class splits / stacks them at the ``state_dict`` boundary.  ``forward_nhwc`` returns (N, H, W, 5 A).
The loss needs per-anchor labels and regression targets (the sampler's output), so a step is

    out = head.train().forward_nhwc(features)
    l_obj, l_box, dout = detection_ops.rpn_loss(out.reshape(-1, 5 * A), labels, targets, A)
    head.backward(features, dout); head.apply_gradients(...)            # input_grad() feeds the backbone

oracle/detection_ref.py (loss) + oracle/mask_head_ref.py-style torch modules (tests) are the checkers.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from .._lib import HOST, check, lib
from ..runtime import as_pointer, torch
from .unet import HipSegmenter, default_init_state


def rpn_head_entries(in_channels, num_anchors, layers=1):
    c, a = in_channels, num_anchors
    ent = []
    for i in range(layers):
        ent += [(f"conv.{i}.0.weight", (c, c, 3, 3), "conv_w"), (f"conv.{i}.0.bias", (c,), "conv_b")]
    return ent + [("head.weight", (5 * a, c, 1, 1), "conv_w"), ("head.bias", (5 * a,), "conv_b")]


class RPNHead(HipSegmenter):
    _first_key = "conv.0.0.weight"

    def __init__(self, in_channels=256, num_anchors=4, layers=1, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (num_anchors, "num_anchors"), (layers, "layers")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if in_channels % 4 or num_anchors % 4:
            raise ValueError("in_channels and num_anchors must be multiples of 4 (16-byte NHWC groups)")
        self.in_channels, self.num_anchors, self.layers = int(in_channels), int(num_anchors), int(layers)
        self.out_channels = 5 * self.num_anchors
        self._entries = rpn_head_entries(self.in_channels, self.num_anchors, self.layers)
        # torch.nn construction order: conv, cls_logits, bbox_pred -- drawn separately, then stacked
        a, c = self.num_anchors, self.in_channels
        split = [e for e in self._entries if not e[0].startswith("head.")] + [
            ("cls_logits.weight", (a, c, 1, 1), "conv_w"), ("cls_logits.bias", (a,), "conv_b"),
            ("bbox_pred.weight", (4 * a, c, 1, 1), "conv_w"), ("bbox_pred.bias", (4 * a,), "conv_b")]
        self._init = self._stack(default_init_state(0, 0, 0, entries=split))
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_rpn_head_create(ctx.handle, self.in_channels, self.layers, self.num_anchors, C.byref(h)))
        return h

    # ---- state_dict with the usual two heads
    @staticmethod
    def _stack(sd):
        if "head.weight" in sd or "cls_logits.weight" not in sd:
            return sd
        out = OrderedDict((k, v) for k, v in sd.items() if not k.startswith(("cls_logits.", "bbox_pred.")))
        t = torch.as_tensor
        out["head.weight"] = torch.cat([t(sd["cls_logits.weight"]), t(sd["bbox_pred.weight"])], 0)
        out["head.bias"] = torch.cat([t(sd["cls_logits.bias"]), t(sd["bbox_pred.bias"])], 0)
        return out

    def state_dict(self):
        sd = super().state_dict()
        a = self.num_anchors
        out = OrderedDict((k, v) for k, v in sd.items() if not k.startswith("head."))
        out["cls_logits.weight"], out["cls_logits.bias"] = sd["head.weight"][:a].clone(), sd["head.bias"][:a].clone()
        out["bbox_pred.weight"], out["bbox_pred.bias"] = sd["head.weight"][a:].clone(), sd["head.bias"][a:].clone()
        return out

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict(self._stack(OrderedDict(state_dict)), strict)

    def grad(self, name):
        a = self.num_anchors
        for pre, sl in (("cls_logits.", slice(0, a)), ("bbox_pred.", slice(a, 5 * a))):
            if name.startswith(pre):
                return super().grad("head." + name[len(pre):])[sl]
        return super().grad(name)

    # ---- backward from the loss kernel's gradient
    def backward(self, features, dout):
        """Parameter gradients (and ``input_grad``) from d(loss)/d(head output); ``features`` is the NHWC input of the
        preceding ``forward_nhwc`` call, ``dout`` (N, H, W, 5 A) or (N H W, 5 A)."""
        n, h, w, c = tuple(features.shape)
        xp, xm, k1 = as_pointer(features, np.float32, self.ctx)
        d = np.ascontiguousarray(np.asarray(dout, np.float32).reshape(n, h, w, self.out_channels))
        check(lib.rfi_model_backward_dlogits(self._h, C.c_void_p(xp), xm, d.ctypes.data_as(C.c_void_p), HOST, n, h, w))
        del k1

    def input_grad(self, shape) -> np.ndarray:
        out = np.empty(tuple(shape), dtype=np.float32)
        check(lib.rfi_model_input_grad(self._h, out.ctypes.data_as(C.c_void_p), HOST))
        return out
