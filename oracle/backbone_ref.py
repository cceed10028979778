"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the ResNet-50-FPN backbone of the Mask R-CNN path (SURVEY.md 8a row A11,
BASELINE.json configs[3]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product (rfi_toolbox_amd) never does.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox contains no detector and torchvision is absent from this image.
The networks are the published ones -- ResNet-50 (He et al. 2016; stride on the 3x3 conv of a Bottleneck) and the Feature
Pyramid Network (Lin et al. 2017) -- with the layer names and the FROZEN BatchNorm (y = (x - running_mean) *
weight / sqrt(running_var + 1e-5) + bias, never updated) of the usual detection backbone, written here as plain torch.nn
modules; the arithmetic is torch's own CPU kernels.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale[None, :, None, None] + shift[None, :, None, None]


class Bottleneck(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = 4 * width
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), FrozenBatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class Body(nn.Module):
    def __init__(self, in_channels, w):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, w, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(w)
        cin = w
        for s, nb in enumerate((3, 4, 6, 3)):
            width = w << s
            blocks = []
            for b in range(nb):
                blocks.append(Bottleneck(cin, width, 2 if (b == 0 and s > 0) else 1))
                cin = 4 * width
            setattr(self, f"layer{s + 1}", nn.Sequential(*blocks))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        out = []
        for s in range(4):
            x = getattr(self, f"layer{s + 1}")(x)
            out.append(x)
        return out


class FPN(nn.Module):
    def __init__(self, in_list, out_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList(nn.Sequential(nn.Conv2d(c, out_channels, 1)) for c in in_list)
        self.layer_blocks = nn.ModuleList(nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, padding=1)) for _ in in_list)

    def forward(self, feats):
        last = self.inner_blocks[-1](feats[-1])
        outs = [self.layer_blocks[-1](last)]
        for i in range(len(feats) - 2, -1, -1):
            lat = self.inner_blocks[i](feats[i])
            last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            outs.insert(0, self.layer_blocks[i](last))
        outs.append(F.max_pool2d(outs[-1], 1, 2, 0))          # LastLevelMaxPool
        return outs


class ResNet50FPN(nn.Module):
    def __init__(self, in_channels=3, base_width=64, fpn_channels=256):
        super().__init__()
        self.body = Body(in_channels, base_width)
        self.fpn = FPN([4 * (base_width << s) for s in range(4)], fpn_channels)

    def forward(self, x):
        return self.fpn(self.body(x))                          # [P2, P3, P4, P5, P6], NCHW


def init_state(in_channels=3, base_width=64, fpn_channels=256, seed=0):
    torch.manual_seed(seed)
    return OrderedDict((k, v.detach().clone()) for k, v in ResNet50FPN(in_channels, base_width, fpn_channels).state_dict().items())
