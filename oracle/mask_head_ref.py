"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-RoI mask branch of Mask R-CNN (BASELINE.json
configs[3]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
(rfi_toolbox_amd) never does.

PARITY UNPINNED BY THE REFERENCE: preshanth/rfi_toolbox contains no detector (placeholder strings only, README.md:90,381,
docs/API.md:180) and torchvision is absent from this image.  The head is the published one (He et al. 2017, fig. 4
right; layer names of torchvision's MaskRCNNHeads / MaskRCNNPredictor):

    mask_fcn1..L: Conv2d(C, C, 3, padding=1) + ReLU;  conv5_mask: ConvTranspose2d(C, C, 2, stride=2) + ReLU;
    mask_fcn_logits: Conv2d(C, K, 1);  loss: mean binary_cross_entropy_with_logits over every RoI pixel.

What IS pinned: the arithmetic is torch's own CPU kernels; the functional ``forward`` below is checked against the
``MaskHeadModule`` nn.Module by tests/test_oracle_golden.py; clip + Adam are the reference's step (oracle/unet_ref.py).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from . import unet_ref


class MaskHeadModule(nn.Module):
    def __init__(self, in_channels=256, num_classes=1, layers=4):
        super().__init__()
        for i in range(1, layers + 1):
            setattr(self, f"mask_fcn{i}", nn.Conv2d(in_channels, in_channels, 3, padding=1))
        self.conv5_mask = nn.ConvTranspose2d(in_channels, in_channels, 2, stride=2)
        self.mask_fcn_logits = nn.Conv2d(in_channels, num_classes, 1)
        self.layers = layers

    def forward(self, x):
        for i in range(1, self.layers + 1):
            x = F.relu(getattr(self, f"mask_fcn{i}")(x))
        return self.mask_fcn_logits(F.relu(self.conv5_mask(x)))


def init_state(in_channels=256, num_classes=1, layers=4, seed=0):
    torch.manual_seed(seed)
    return OrderedDict((k, v.detach().clone()) for k, v in MaskHeadModule(in_channels, num_classes, layers).state_dict().items())


def forward(state, x_nchw, training=False, buffer_updates=None, tape=None):
    h, i = x_nchw, 1
    while f"mask_fcn{i}.weight" in state:
        h = torch.relu(unet_ref._conv3x3(h, state[f"mask_fcn{i}.weight"], state[f"mask_fcn{i}.bias"]))
        i += 1
    h = torch.relu(unet_ref._convt2x2(h, state["conv5_mask.weight"], state["conv5_mask.bias"]))
    return F.conv2d(h, state["mask_fcn_logits.weight"], state["mask_fcn_logits.bias"])


def mask_loss(logits, target):
    return F.binary_cross_entropy_with_logits(logits.reshape(-1), target.reshape(-1).to(logits.dtype))


def loss_and_grads(state, x_nchw, y):
    """-> loss, logits, parameter gradients, gradient w.r.t. the input features."""
    x = x_nchw.detach().clone().requires_grad_(True)
    names = unet_ref.param_names(state)
    work = OrderedDict(state)
    leaves = []
    for k in names:
        t = state[k].detach().clone().requires_grad_(True)
        work[k] = t
        leaves.append(t)
    logits = forward(work, x)
    loss = mask_loss(logits, y)
    grads = torch.autograd.grad(loss, leaves + [x])
    return loss.detach(), logits.detach(), OrderedDict(zip(names, grads[:-1])), grads[-1]


def train_step(state, adam, x_nchw, y, **kw):
    return unet_ref.train_step(state, adam, x_nchw, y, forward_fn=forward, loss_fn=mask_loss, **kw)


# ---------------------------------------------------------------- RPN head (Faster R-CNN; same status: builder-defined)
class RPNHeadModule(nn.Module):
    """conv (3x3 + ReLU) x layers, cls_logits (A), bbox_pred (4 A, anchor-major) -- torchvision's RPNHead layer names."""

    def __init__(self, in_channels=256, num_anchors=4, layers=1):
        super().__init__()
        self.conv = nn.Sequential(*[nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.ReLU()) for _ in range(layers)])
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, 4 * num_anchors, 1)

    def forward(self, x):
        t = self.conv(x)
        return self.cls_logits(t), self.bbox_pred(t)


def rpn_init_state(in_channels=256, num_anchors=4, layers=1, seed=0):
    torch.manual_seed(seed)
    return OrderedDict((k, v.detach().clone()) for k, v in RPNHeadModule(in_channels, num_anchors, layers).state_dict().items())


def rpn_forward(state, x_nchw):
    """-> head output (N, H, W, 5 A): A objectness logits then A x 4 deltas per pixel (the library's layout)."""
    h, i = x_nchw, 0
    while f"conv.{i}.0.weight" in state:
        h = torch.relu(unet_ref._conv3x3(h, state[f"conv.{i}.0.weight"], state[f"conv.{i}.0.bias"]))
        i += 1
    cls = F.conv2d(h, state["cls_logits.weight"], state["cls_logits.bias"])
    box = F.conv2d(h, state["bbox_pred.weight"], state["bbox_pred.bias"])
    return torch.cat([cls, box], 1).permute(0, 2, 3, 1)


def rpn_loss_torch(out, labels, targets, A, beta=1.0 / 9):
    """The RPN loss (objectness BCE over sampled anchors + smooth L1 over positives, both / num_sampled) in torch ops."""
    P = out.shape[0] * out.shape[1] * out.shape[2]
    o = out.reshape(P, 5 * A)
    lab = torch.as_tensor(labels).reshape(P, A)
    tgt = torch.as_tensor(targets, dtype=o.dtype).reshape(P, A, 4)
    n = max(int((lab >= 0).sum()), 1)
    x, d = o[:, :A], o[:, A:].reshape(P, A, 4)
    samp, pos = lab >= 0, lab > 0
    l_obj = F.binary_cross_entropy_with_logits(x[samp], pos[samp].to(o.dtype), reduction="sum") / n
    l_box = F.smooth_l1_loss(d[pos], tgt[pos], beta=beta, reduction="sum") / n
    return l_obj, l_box


# ---------------------------------------------------------------- box head (TwoMLPHead + FastRCNNPredictor; builder-defined)
class BoxHeadModule(nn.Module):
    def __init__(self, in_channels=256, resolution=7, representation_size=1024, num_classes=2):
        super().__init__()
        self.fc6 = nn.Linear(in_channels * resolution ** 2, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)
        self.cls_score = nn.Linear(representation_size, num_classes)
        self.bbox_pred = nn.Linear(representation_size, 4 * num_classes)

    def forward(self, x_nchw):                                    # (R, C, res, res)
        x = F.relu(self.fc7(F.relu(self.fc6(x_nchw.flatten(1)))))
        return self.cls_score(x), self.bbox_pred(x)


def fastrcnn_loss_torch(cls, box, labels, targets, beta=1.0 / 9):
    labels = torch.as_tensor(labels, dtype=torch.long)
    tgt = torch.as_tensor(targets, dtype=box.dtype)
    l_cls = F.cross_entropy(cls, labels)
    pos = torch.where(labels > 0)[0]
    b = box.reshape(box.shape[0], -1, 4)
    l_box = F.smooth_l1_loss(b[pos, labels[pos]], tgt[pos], beta=beta, reduction="sum") / labels.numel()
    return l_cls, l_box
