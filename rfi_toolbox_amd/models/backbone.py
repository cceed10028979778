"""``ResNet50FPN`` -- the backbone of the Mask R-CNN path (BASELINE.json configs[3]).

Not in the reference (no detector there) and torchvision is absent: builder-defined as the published networks (ResNet-50
with the stride on the 3x3 conv of a Bottleneck; Feature Pyramid Network; ``LastLevelMaxPool``) with the layer names and
the frozen BatchNorm of the usual detection backbone (``body.*``, ``fpn.inner_blocks.*``, ``fpn.layer_blocks.*``).
oracle/backbone_ref.py holds the same networks as plain ``torch.nn`` modules (parity unpinned by the reference).

    feats = backbone.forward_features(images_nhwc)            # [P2, P3, P4, P5, P6], NHWC, strides 4 .. 64
    ...                                                       # heads + losses produce d(loss)/d(P_i)
    backbone.backward(images_nhwc, dfeats); backbone.apply_gradients(...)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import HOST, check, lib
from ..runtime import as_pointer
from .unet import HipSegmenter, default_init_state


def _bn(p, c):
    return [(f"{p}.weight", (c,), "bn_g"), (f"{p}.bias", (c,), "bn_b"), (f"{p}.running_mean", (c,), "bn_rm"),
            (f"{p}.running_var", (c,), "bn_rv")]


def resnet50_fpn_entries(in_channels, base_width, fpn_channels):
    w, f = base_width, fpn_channels
    ent = [("body.conv1.weight", (w, in_channels, 7, 7), "conv_w")] + _bn("body.bn1", w)
    cin = w
    for s, nb in enumerate((3, 4, 6, 3)):
        width = w << s
        for b in range(nb):
            p = f"body.layer{s + 1}.{b}"
            ent += [(f"{p}.conv1.weight", (width, cin, 1, 1), "conv_w")] + _bn(f"{p}.bn1", width)
            ent += [(f"{p}.conv2.weight", (width, width, 3, 3), "conv_w")] + _bn(f"{p}.bn2", width)
            ent += [(f"{p}.conv3.weight", (4 * width, width, 1, 1), "conv_w")] + _bn(f"{p}.bn3", 4 * width)
            if b == 0:
                ent += [(f"{p}.downsample.0.weight", (4 * width, cin, 1, 1), "conv_w")] + _bn(f"{p}.downsample.1", 4 * width)
            cin = 4 * width
    for i in range(4):
        ent += [(f"fpn.inner_blocks.{i}.0.weight", (f, 4 * (w << i), 1, 1), "conv_w"), (f"fpn.inner_blocks.{i}.0.bias", (f,), "conv_b")]
    for i in range(4):
        ent += [(f"fpn.layer_blocks.{i}.0.weight", (f, f, 3, 3), "conv_w"), (f"fpn.layer_blocks.{i}.0.bias", (f,), "conv_b")]
    return ent


class ResNet50FPN(HipSegmenter):
    _first_key = "body.conv1.weight"

    def __init__(self, in_channels=3, base_width=64, fpn_channels=256, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (base_width, "base_width"), (fpn_channels, "fpn_channels")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if base_width % 4 or fpn_channels % 4:
            raise ValueError("base_width and fpn_channels must be multiples of 4")
        self.in_channels, self.base_width, self.out_channels = int(in_channels), int(base_width), int(fpn_channels)
        self._entries = resnet50_fpn_entries(self.in_channels, self.base_width, self.out_channels)
        self._init = default_init_state(0, 0, 0, entries=self._entries)
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_resnet50_fpn_create(ctx.handle, self.in_channels, self.base_width, self.out_channels, C.byref(h)))
        return h

    def _shapes(self, n, h, w):
        return [(n, h >> (2 + i), w >> (2 + i), self.out_channels) for i in range(5)]

    def forward_features(self, x):
        """x (N, H, W, C) float32, H and W multiples of 64 -> [P2, P3, P4, P5, P6] as NumPy NHWC arrays."""
        n, h, w, c = tuple(x.shape)
        if c != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {c}")
        xp, xm, keep = as_pointer(x, np.float32, self.ctx)
        outs = [np.empty(s, np.float32) for s in self._shapes(n, h, w)]
        ptrs = (C.c_void_p * 5)(*[o.ctypes.data for o in outs])
        check(lib.rfi_backbone_forward(self._h, C.c_void_p(xp), xm, n, h, w, ptrs, HOST))
        del keep
        return outs

    def backward(self, x, dfeats):
        """Parameter gradients from d(loss)/d(P_i) (``None`` entries count as zero) after ``forward_features(x)``."""
        n, h, w, _ = tuple(x.shape)
        xp, xm, keep = as_pointer(x, np.float32, self.ctx)
        arrs = [None if d is None else np.ascontiguousarray(np.asarray(d, np.float32).reshape(s))
                for d, s in zip(dfeats, self._shapes(n, h, w))]
        ptrs = (C.c_void_p * 5)(*[None if a is None else a.ctypes.data for a in arrs])
        check(lib.rfi_backbone_backward(self._h, C.c_void_p(xp), xm, n, h, w, ptrs, HOST))
        del keep
