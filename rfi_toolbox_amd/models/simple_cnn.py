"""``SimpleCNN`` -- the "3-layer CNN segmenter" of BASELINE.json configs[0]/[1].

The reference ships no such class; the nearest text is the elided "custom model" example of its
README (README.md:379-398: ``encoder = Sequential(Conv2d(3, 64, 3, padding=1), ReLU(), ...)``,
``decoder = Sequential(..., Conv2d(64, 1, 1), ...)``).  This build fixes it as

    encoder = Sequential(Conv2d(in, 64, 3, padding=1), ReLU(), Conv2d(64, 64, 3, padding=1), ReLU())
    decoder = Sequential(Conv2d(64, out, 1))                     # logits, like UNet.forward

trained with the same step as the U-Net (BCE-with-logits + dice, clip 1.0, Adam;
scripts/train_model.py:120-151).  ``state_dict`` keys are those ``torch.nn`` gives that module:
``encoder.0.weight/bias``, ``encoder.2.weight/bias``, ``decoder.0.weight/bias``.  Parity for this
model is builder-defined (no reference implementation exists); the oracle is the same three
``torch.nn`` layers (oracle/cnn_ref.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import check, lib
from .unet import HipSegmenter, default_init_state


def simple_cnn_entries(in_channels, out_channels, width):
    return [("encoder.0.weight", (width, in_channels, 3, 3), "conv_w"), ("encoder.0.bias", (width,), "conv_b"),
            ("encoder.2.weight", (width, width, 3, 3), "conv_w"), ("encoder.2.bias", (width,), "conv_b"),
            ("decoder.0.weight", (out_channels, width, 1, 1), "conv_w"), ("decoder.0.bias", (out_channels,), "conv_b")]


class SimpleCNN(HipSegmenter):
    _first_key = "encoder.0.weight"

    def __init__(self, in_channels=3, out_channels=1, width=64, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (out_channels, "out_channels"), (width, "width")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        self.in_channels, self.out_channels, self.width = int(in_channels), int(out_channels), int(width)
        self._entries = simple_cnn_entries(self.in_channels, self.out_channels, self.width)
        # same draws, same order as constructing the torch.nn module under torch.manual_seed
        self._init = default_init_state(0, 0, 0, entries=self._entries)
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_cnn3_create(ctx.handle, self.in_channels, self.out_channels, self.width, C.byref(h)))
        return h
