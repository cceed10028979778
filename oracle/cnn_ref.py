"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the "3-layer CNN segmenter" (SURVEY.md 8a row A9,
BASELINE.json configs[0]/[1]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; the product (rfi_toolbox_amd) never does.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox ships no such model (nearest text: the
elided example README.md:379-398).  The build defines it as

    Conv2d(in, C, 3, padding=1) -> ReLU -> Conv2d(C, C, 3, padding=1) -> ReLU -> Conv2d(C, out, 1)

returning logits, with ``torch.nn`` state_dict keys ``encoder.0.*``, ``encoder.2.*``, ``decoder.0.*``.
The arithmetic is torch's own CPU kernels (F.conv2d / relu / autograd), pinned against an
``nn.Sequential`` of the same layers by tests/golden/make_golden.py (fixture cnn3_c16_b4_s32.npz).
Loss, clipping and Adam are the reference's step (scripts/train_model.py:120-151), shared with
oracle/unet_ref.py.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import unet_ref


def entries(in_channels=3, out_channels=1, width=64):
    return [("encoder.0.weight", (width, in_channels, 3, 3)), ("encoder.0.bias", (width,)),
            ("encoder.2.weight", (width, width, 3, 3)), ("encoder.2.bias", (width,)),
            ("decoder.0.weight", (out_channels, width, 1, 1)), ("decoder.0.bias", (out_channels,))]


def init_state(in_channels=3, out_channels=1, width=64, seed=0):
    """Default torch.nn.Conv2d initialisation drawn in module-construction order."""
    torch.manual_seed(seed)
    enc = torch.nn.Sequential(torch.nn.Conv2d(in_channels, width, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(width, width, 3, padding=1), torch.nn.ReLU())
    dec = torch.nn.Sequential(torch.nn.Conv2d(width, out_channels, 1))
    st = OrderedDict()
    for k, v in enc.state_dict().items():
        st[f"encoder.{k}"] = v.detach().clone()
    for k, v in dec.state_dict().items():
        st[f"decoder.{k}"] = v.detach().clone()
    return st


def forward(state, x_nchw, training=False, buffer_updates=None, tape=None):
    h = F.conv2d(x_nchw, state["encoder.0.weight"], state["encoder.0.bias"], padding=1)
    if tape is not None:
        tape["encoder.0.out"] = h
    h = F.relu(h)
    h = F.conv2d(h, state["encoder.2.weight"], state["encoder.2.bias"], padding=1)
    if tape is not None:
        tape["encoder.2.out"] = h
    h = F.relu(h)
    return F.conv2d(h, state["decoder.0.weight"], state["decoder.0.bias"])


def loss_and_grads(state, x_nchw, y):
    return unet_ref.loss_and_grads(state, x_nchw, y, training=True, forward_fn=forward)


def train_step(state, adam, x_nchw, y, **kw):
    return unet_ref.train_step(state, adam, x_nchw, y, forward_fn=forward, **kw)


new_adam_state = unet_ref.new_adam_state
