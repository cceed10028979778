"""``MaskHead`` -- the per-RoI mask branch of Mask R-CNN (BASELINE.json configs[3], north_star "per-pixel mask head";
SURVEY.md 8a row A11).

The reference contains no detector and torchvision is absent from this image, so the class is builder-defined as the
published head (He et al. 2017, fig. 4 right; the layer names are those of torchvision's ``MaskRCNNHeads`` /
``MaskRCNNPredictor``):

    mask_fcn1..L     = Conv2d(C, C, 3, padding=1) -> ReLU                      # L = 4
    conv5_mask       = ConvTranspose2d(C, C, 2, stride=2) -> ReLU
    mask_fcn_logits  = Conv2d(C, num_classes, 1)

on RoIAlign-ed features ``(R, 14, 14, C)`` (``detection_ops.roi_align``) -> logits ``(R, 28, 28, num_classes)``; the
training step is the U-Net's (clip + Adam) with the mean binary cross-entropy over every RoI pixel as the loss
(``num_classes == 1``: one foreground class, RFI).  ``input_grad()`` returns the gradient w.r.t. the RoI features of
the last backward pass, which ``detection_ops.roi_align_backward`` scatters into the feature map.
oracle/mask_head_ref.py holds the same layers as plain ``torch.nn`` modules (parity unpinned by the reference).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import HOST, check, lib
from .unet import HipSegmenter, default_init_state


def mask_head_entries(in_channels, num_classes, layers=4):
    c = in_channels
    ent = []
    for i in range(1, layers + 1):
        ent += [(f"mask_fcn{i}.weight", (c, c, 3, 3), "conv_w"), (f"mask_fcn{i}.bias", (c,), "conv_b")]
    ent += [("conv5_mask.weight", (c, c, 2, 2), "conv_w"), ("conv5_mask.bias", (c,), "conv_b"),
            ("mask_fcn_logits.weight", (num_classes, c, 1, 1), "conv_w"), ("mask_fcn_logits.bias", (num_classes,), "conv_b")]
    return ent


class MaskHead(HipSegmenter):
    _first_key = "mask_fcn1.weight"
    _out_scale = 2

    def __init__(self, in_channels=256, num_classes=1, layers=4, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (num_classes, "num_classes"), (layers, "layers")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if in_channels % 4:
            raise ValueError(f"in_channels must be a multiple of 4, got {in_channels}")
        self.in_channels, self.out_channels, self.layers = int(in_channels), int(num_classes), int(layers)
        self._entries = mask_head_entries(self.in_channels, self.out_channels, self.layers)
        # same draws, same order as constructing the torch.nn modules under torch.manual_seed
        self._init = default_init_state(0, 0, 0, entries=self._entries)
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_mask_head_create(ctx.handle, self.in_channels, self.layers, self.out_channels, C.byref(h)))
        return h

    def input_grad(self, shape) -> np.ndarray:
        """Gradient w.r.t. the RoI features of the last ``forward_backward`` / ``train_step``; ``shape`` = (R, h, w, C)."""
        out = np.empty(tuple(shape), dtype=np.float32)
        check(lib.rfi_model_input_grad(self._h, out.ctypes.data_as(C.c_void_p), HOST))
        return out
