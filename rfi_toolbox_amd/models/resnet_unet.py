"""``UNetResNet18`` -- the "U-Net with a ResNet-18 encoder" of BASELINE.json configs[2].

The reference ships plain U-Nets only (rfi_toolbox/models/unet.py) and this image holds neither torchvision nor
segmentation_models_pytorch, so the class is builder-defined:

    stem    = Sequential(Conv2d(in, f, 3, padding=1, bias=False), BatchNorm2d(f), ReLU())        # full resolution
    layer l = Sequential(BasicBlock(c_{l-1}, c_l, stride s_l), BasicBlock(c_l, c_l, 1))           # c_l = f 2^(l-1)
    bottleneck / decoder4..1 / final_conv: the reference's own (models/unet.py:30-77), skips = the four stage outputs

with ``BasicBlock`` the published ResNet-18 block (conv3x3 - BN - ReLU - conv3x3 - BN, identity or Conv1x1(stride 2)
+ BN shortcut, ReLU).  ``state_dict`` keys are those ``torch.nn`` gives that module tree (oracle/resnet_unet_ref.py
holds it as plain ``torch.nn`` modules).  The optimisation step is the U-Net's (scripts/train_model.py:120-151).

``set_compute_dtype("bfloat16")`` selects the bfloat16 data flow (csrc/model_planes.cpp: every tensor between kernels in HBM
as bfloat16, the stride-2 3x3 / 1x1 layers as strided plane contractions, their input gradient by parity classes) when
``init_features % 16 == 0``; other widths keep float32 tensors and round the operands at staging.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import check, lib
from .unet import HipSegmenter, _double_conv_entries, default_init_state


def _bn_entries(p, c):
    return [(f"{p}.weight", (c,), "bn_g"), (f"{p}.bias", (c,), "bn_b"), (f"{p}.running_mean", (c,), "bn_rm"),
            (f"{p}.running_var", (c,), "bn_rv"), (f"{p}.num_batches_tracked", (), "bn_nbt")]


def resnet_unet_entries(in_channels, out_channels, init_features):
    f = init_features
    ent = [("stem.0.weight", (f, in_channels, 3, 3), "conv_w")] + _bn_entries("stem.1", f)
    cin = f
    for lvl in range(1, 5):
        cout = f << (lvl - 1)
        for b in range(2):
            p = f"layer{lvl}.{b}"
            ent += [(f"{p}.conv1.weight", (cout, cin, 3, 3), "conv_w")] + _bn_entries(f"{p}.bn1", cout)
            ent += [(f"{p}.conv2.weight", (cout, cout, 3, 3), "conv_w")] + _bn_entries(f"{p}.bn2", cout)
            if b == 0 and lvl > 1:
                ent += [(f"{p}.downsample.0.weight", (cout, cin, 1, 1), "conv_w")] + _bn_entries(f"{p}.downsample.1", cout)
            cin = cout
    ent += _double_conv_entries("bottleneck.conv", cin, 2 * cin)
    cin *= 2
    for lvl in range(4, 0, -1):
        cout = f << (lvl - 1)
        ent += [(f"decoder{lvl}.up.weight", (cin, cout, 2, 2), "conv_w"), (f"decoder{lvl}.up.bias", (cout,), "conv_b")]
        ent += _double_conv_entries(f"decoder{lvl}.conv.conv", cin, cout)
        cin = cout
    ent += [("final_conv.weight", (out_channels, f, 1, 1), "conv_w"), ("final_conv.bias", (out_channels,), "conv_b")]
    return ent


class UNetResNet18(HipSegmenter):
    _first_key = "stem.0.weight"

    def __init__(self, in_channels=3, out_channels=1, init_features=64, *, device=None):
        for v, nm in ((in_channels, "in_channels"), (out_channels, "out_channels"), (init_features, "init_features")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if init_features % 4:
            raise ValueError(f"init_features must be a multiple of 4, got {init_features}")
        self.in_channels, self.out_channels, self.init_features = int(in_channels), int(out_channels), int(init_features)
        self.depth = 4
        self._entries = resnet_unet_entries(self.in_channels, self.out_channels, self.init_features)
        # same draws, same order as constructing the torch.nn module tree under torch.manual_seed
        self._init = default_init_state(0, 0, 0, entries=self._entries)
        self._setup(device)

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_unet_resnet_create(ctx.handle, self.in_channels, self.out_channels, self.init_features, C.byref(h)))
        return h
