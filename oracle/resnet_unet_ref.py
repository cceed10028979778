"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the "U-Net with a ResNet-18-style encoder" (SURVEY.md 8a row A10,
BASELINE.json configs[2]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this;
the product (rfi_toolbox_amd) never does.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox ships plain U-Nets only (models/unet.py) and neither
torchvision nor segmentation_models_pytorch exists in this image, so the model is builder-defined:

    stem      Conv2d(in, f, 3, padding=1, bias=False) + BatchNorm2d + ReLU                       (full resolution)
    layer l   BasicBlock(c_{l-1} -> c_l, stride s_l), BasicBlock(c_l -> c_l);  c_l = f 2^(l-1), s_1 = 1, s_{2..4} = 2
              BasicBlock = conv3x3(stride) - BN - ReLU - conv3x3 - BN, + identity or Conv1x1(stride 2) - BN, ReLU
              (He et al. 2016, fig. 5 left / torchvision.models.resnet.BasicBlock's published structure)
    then      the reference's own bottleneck (MaxPool2d(2) + DoubleConv), DecoderBlocks and 1x1 head
              (models/unet.py:7-39, :52-77) with the four stage outputs as skip connections.

What IS pinned: the arithmetic is torch's own CPU kernels; ``forward`` below (functional, on a state_dict) is
checked against ``ResNetUNet`` -- the same graph written as ordinary ``torch.nn`` modules -- by
tests/test_oracle_golden.py, and its decoder half is the reference-pinned oracle/unet_ref.py code.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from . import unet_ref


# ------------------------------------------------------------------ the graph as torch.nn modules
class _DoubleConv(nn.Module):                       # models/unet.py:7-18
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                                  nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class _DecoderBlock(nn.Module):                     # models/unet.py:30-39
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2)
        self.conv = _DoubleConv(cin, cout)

    def forward(self, x, skip):
        return self.conv(torch.cat([self.up(x), skip], dim=1))


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNetUNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1, init_features=64):
        super().__init__()
        f = init_features
        self.stem = nn.Sequential(nn.Conv2d(in_channels, f, 3, padding=1, bias=False), nn.BatchNorm2d(f), nn.ReLU(inplace=True))
        cin = f
        for lvl in range(1, 5):
            cout = f << (lvl - 1)
            setattr(self, f"layer{lvl}", nn.Sequential(BasicBlock(cin, cout, 1 if lvl == 1 else 2), BasicBlock(cout, cout, 1)))
            cin = cout
        self.pool = nn.MaxPool2d(2)
        self.bottleneck = _DoubleConv(cin, 2 * cin)
        cin *= 2
        for lvl in range(4, 0, -1):
            cout = f << (lvl - 1)
            setattr(self, f"decoder{lvl}", _DecoderBlock(cin, cout))
            cin = cout
        self.final_conv = nn.Conv2d(f, out_channels, 1)

    def forward(self, x):
        h = self.stem(x)
        skips = []
        for lvl in range(1, 5):
            h = getattr(self, f"layer{lvl}")(h)
            skips.append(h)
        h = self.bottleneck(self.pool(h))
        for lvl in range(4, 0, -1):
            h = getattr(self, f"decoder{lvl}")(h, skips[lvl - 1])
        return self.final_conv(h)


def init_state(in_channels=3, out_channels=1, init_features=64, seed=0):
    """torch's default initialisation, drawn in module-construction order."""
    torch.manual_seed(seed)
    return OrderedDict((k, v.detach().clone()) for k, v in ResNetUNet(in_channels, out_channels, init_features).state_dict().items())


def entries(in_channels=3, out_channels=1, init_features=64):
    with torch.device("meta"):
        sd = ResNetUNet(in_channels, out_channels, init_features).state_dict()
    return [(k, tuple(v.shape)) for k, v in sd.items()]


# ------------------------------------------------------------------ functional forward on a state_dict
class _ConvBF16S(torch.autograd.Function):
    """bf16-operand convolution with a stride (see unet_ref._ConvBF16).  ``unet_ref.bf16_operands(round_outputs=True)`` (the
    library's bfloat16 data flow, widths in whole 16-channel chunks): the output and -- ``round_dx`` -- the input gradient
    are stored as bfloat16 tensors."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, round_dx=False):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, bool(round_dx))
        y = F.conv2d(unet_ref._bf(x), unet_ref._bf(w), None, stride=stride, padding=padding)
        return unet_ref._bf(y) if _storage16() else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, round_dx = ctx.cfg
        dyb = unet_ref._bf(dy)
        dx = torch.nn.grad.conv2d_input(x.shape, unet_ref._bf(w), dyb, stride=stride, padding=padding)
        dw = torch.nn.grad.conv2d_weight(unet_ref._bf(x), w.shape, dyb, stride=stride, padding=padding)
        return (unet_ref._bf(dx) if round_dx else dx), dw, None, None, None


def _storage16():
    """The bfloat16 data flow of the library's ResNet-encoder model: ``bf16_operands(round_outputs=True)`` and a width in
    whole 16-channel chunks (else the library keeps float32 tensors and rounds operands only)."""
    return unet_ref._BF16_OPERANDS and unet_ref._BF16_ROUND_OUTPUTS and _WIDTH_16


_WIDTH_16 = False           # set by forward()


class _StoreBF16(torch.autograd.Function):
    """A tensor AND its gradient stored as bfloat16 (activations between blocks; the sums of gradient terms that meet there)."""

    @staticmethod
    def forward(ctx, x):
        return unet_ref._bf(x)

    @staticmethod
    def backward(ctx, g):
        return unet_ref._bf(g)


def _conv(x, w, stride, padding, round_dx=False):
    if unet_ref._BF16_OPERANDS:
        return _ConvBF16S.apply(x, w, stride, padding, round_dx and _storage16())
    return F.conv2d(x, w, None, stride=stride, padding=padding)


def _block(h, st, p, training, bu, tape):
    stride = 2 if f"{p}.downsample.0.weight" in st else 1
    # bfloat16 flow: conv2's input gradient and conv1's are bfloat16 tensors; the block input's gradient (their sum with the
    # shortcut's) is rounded where it is masked by the previous block's ReLU; the output is a bfloat16 tensor
    y = _conv(h, st[f"{p}.conv1.weight"], stride, 1, round_dx=True)
    if tape is not None:
        tape[f"{p}.conv1.out"] = y
    y = torch.relu(unet_ref._bn(y, st, f"{p}.bn1", training, 1, bu, tape, f"{p}.bn1"))
    y = _conv(y, st[f"{p}.conv2.weight"], 1, 1, round_dx=True)
    if tape is not None:
        tape[f"{p}.conv2.out"] = y
    y = unet_ref._bn(y, st, f"{p}.bn2", training, 1, bu, tape, f"{p}.bn2")
    if stride == 2:
        s = _conv(h, st[f"{p}.downsample.0.weight"], 2, 0)       # (its input gradient leaves inside conv1's: one bfloat16 sum)
        if tape is not None:
            tape[f"{p}.downsample.0.out"] = s
        h = unet_ref._bn(s, st, f"{p}.downsample.1", training, 1, bu, tape, f"{p}.downsample.1")
    z = y + h
    if _storage16():
        z = _StoreBF16.apply(z)       # (rounding commutes with the ReLU and its mask: out and dz are the stored tensors)
    out = torch.relu(z)
    if tape is not None:
        tape[f"{p}.out"] = out
    return out


def forward(state, x_nchw, training=False, buffer_updates=None, tape=None):
    global _WIDTH_16
    st, bu = state, buffer_updates
    _WIDTH_16 = st["stem.0.weight"].shape[0] % 16 == 0
    unet_ref._WIDTHS_16 = _WIDTH_16               # (the decoder half below is unet_ref's)
    unet_ref._WIDTHS_32 = st["stem.0.weight"].shape[0] % 32 == 0
    h = _conv(x_nchw, st["stem.0.weight"], 1, 1)
    if tape is not None:
        tape["stem.0.out"] = h
    h = torch.relu(unet_ref._bn(h, st, "stem.1", training, 1, bu, tape, "stem.1"))
    if _storage16():
        h = _StoreBF16.apply(h)
    skips = []
    for lvl in range(1, 5):
        for b in range(2):
            h = _block(h, st, f"layer{lvl}.{b}", training, bu, tape)
        skips.append(h)
    h = F.max_pool2d(h, kernel_size=2, stride=2)
    h = unet_ref._double_conv(h, st, "bottleneck.conv", training, 1, bu, tape)
    for lvl in range(4, 0, -1):
        up = unet_ref._convt2x2(h, st[f"decoder{lvl}.up.weight"], st[f"decoder{lvl}.up.bias"])
        h = torch.cat([up, skips[lvl - 1]], dim=1)
        h = unet_ref._double_conv(h, st, f"decoder{lvl}.conv.conv", training, 1, bu, tape)
    if _storage16() and unet_ref._BF16_ROUND_GRADS and h.requires_grad and st["final_conv.weight"].shape[0] == 1:
        h.register_hook(unet_ref._bf)     # the gradient the 1x1 head sends into the last DoubleConv (as unet_ref.forward)
    return F.conv2d(h, st["final_conv.weight"], st["final_conv.bias"])


def loss_and_grads(state, x_nchw, y, tape=None):
    return unet_ref.loss_and_grads(state, x_nchw, y, training=True, tape=tape, forward_fn=forward)


def train_step(state, adam, x_nchw, y, **kw):
    return unet_ref.train_step(state, adam, x_nchw, y, forward_fn=forward, **kw)
