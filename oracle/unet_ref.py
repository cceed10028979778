"""torch-CPU fp32 oracle for the U-Net training path.  TEST INFRASTRUCTURE ONLY.

Restates, as flat functions over a ``{name: tensor}`` state (names are the
reference's ``state_dict`` keys), what these reference pieces compute:

* ``DoubleConv``  Conv3x3(p=1)+BN+ReLU twice      rfi_toolbox/models/unet.py:6-19
* ``Encoder``     ``pool(conv(x)), conv(x)``       rfi_toolbox/models/unet.py:21-28
* ``Decoder``     ConvT(k2,s2) -> cat([up,skip])   rfi_toolbox/models/unet.py:30-39
* ``UNet``        4 levels + bottleneck + 1x1 head rfi_toolbox/models/unet.py:41-77
* loss            BCEWithLogits + dice             rfi_toolbox/scripts/train_model.py:120-128,146
* step            clip_grad_norm_(1.0) + Adam(L2)  rfi_toolbox/scripts/train_model.py:130,142-151

Deliberate restatement choices (all pinned by tests/golden):

* The reference's ``Encoder.forward`` evaluates its DoubleConv twice on the
  same input (unet.py:28).  In train mode both evaluations see the same batch
  statistics, so the values are identical; the only observable side effect is
  that the encoder BatchNorm running statistics receive the EMA update twice
  and ``num_batches_tracked`` advances by 2.  Here each block is evaluated
  once and ``ema_repeats=2`` is applied for encoder blocks.
* BatchNorm is written out (batch mean, biased variance for normalisation,
  unbiased variance for the running estimate, momentum 0.1, eps 1e-5, closed-form
  backward) instead of calling ``torch.nn.BatchNorm2d``.
* Adam is written out from the formula torch.optim.Adam implements (coupled
  L2: ``g += wd * p``; bias-corrected; ``denom = sqrt(v)/sqrt(bc2) + eps``).

Backward uses ``torch.autograd`` exactly as the reference does
(``loss.backward()``, train_model.py:148); on CPU the reference's autocast /
GradScaler are disabled (train_model.py:131,144) so the path is pure fp32.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1
BN_EPS = 1e-5


# --------------------------------------------------------------------------
# parameter table
# --------------------------------------------------------------------------
def _double_conv_entries(prefix, cin, cout):
    out = []
    for conv_idx, bn_idx, ci in ((0, 1, cin), (3, 4, cout)):
        out.append((f"{prefix}.{conv_idx}.weight", (cout, ci, 3, 3), "param"))
        out.append((f"{prefix}.{conv_idx}.bias", (cout,), "param"))
        out.append((f"{prefix}.{bn_idx}.weight", (cout,), "param"))
        out.append((f"{prefix}.{bn_idx}.bias", (cout,), "param"))
        out.append((f"{prefix}.{bn_idx}.running_mean", (cout,), "buffer"))
        out.append((f"{prefix}.{bn_idx}.running_var", (cout,), "buffer"))
        out.append((f"{prefix}.{bn_idx}.num_batches_tracked", (), "counter"))
    return out


def unet_entries(in_channels=1, out_channels=1, init_features=32, depth=4):
    """Ordered (name, shape, kind) list == reference ``UNet.state_dict()`` order."""
    f = init_features
    ent = []
    cin = in_channels
    for lvl in range(1, depth + 1):
        cout = f * 2 ** (lvl - 1)
        ent += _double_conv_entries(f"encoder{lvl}.conv.conv", cin, cout)
        cin = cout
    ent += _double_conv_entries("bottleneck.conv", cin, cin * 2)
    cin = cin * 2
    for lvl in range(depth, 0, -1):
        cout = f * 2 ** (lvl - 1)
        ent.append((f"decoder{lvl}.up.weight", (cin, cout, 2, 2), "param"))
        ent.append((f"decoder{lvl}.up.bias", (cout,), "param"))
        ent += _double_conv_entries(f"decoder{lvl}.conv.conv", cin, cout)
        cin = cout
    ent.append(("final_conv.weight", (out_channels, f, 1, 1), "param"))
    ent.append(("final_conv.bias", (out_channels,), "param"))
    return ent


def init_state(in_channels=1, out_channels=1, init_features=32, depth=4, seed=0):
    """Random state with torch's default init *distributions* (not its RNG stream)."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    for name, shape, kind in unet_entries(in_channels, out_channels, init_features, depth):
        if kind == "counter":
            st[name] = torch.zeros((), dtype=torch.int64)
        elif kind == "buffer":
            st[name] = torch.ones(shape) if name.endswith("running_var") else torch.zeros(shape)
        elif len(shape) == 4:                       # conv / convT weight
            fan_in = shape[1] * shape[2] * shape[3]
            if ".up." in name:                      # ConvTranspose2d: fan_in uses dim 1 too
                fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            st[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif name.endswith(".bias") and (".0.bias" in name or ".3.bias" in name
                                         or ".up.bias" in name or name == "final_conv.bias"):
            wname = name[:-4] + "weight"
            ws = st[wname].shape
            b = 1.0 / math.sqrt(ws[1] * ws[2] * ws[3])
            st[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif name.endswith(".weight"):              # BN gamma
            st[name] = torch.ones(shape)
        else:                                       # BN beta
            st[name] = torch.zeros(shape)
    return st


def param_names(state):
    return [k for k, v in state.items()
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))]


def infer_config(state):
    f = state["encoder1.conv.conv.0.weight"].shape[0]
    cin = state["encoder1.conv.conv.0.weight"].shape[1]
    cout = state["final_conv.weight"].shape[0]
    depth = 0
    while f"encoder{depth + 1}.conv.conv.0.weight" in state:
        depth += 1
    return cin, cout, f, depth


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
class _BatchNormTrain(torch.autograd.Function):
    """Train-mode BatchNorm2d with the closed-form backward

        dx = gamma*invstd * (dy - mean(dy) - xhat * mean(dy*xhat))
        dgamma = sum(dy*xhat),  dbeta = sum(dy)

    Differentiating the textbook forward formula op by op through autograd loses
    ~3 decimal digits to cancellation on small batches; torch's own BatchNorm
    (what the reference runs) uses this closed form, and so do the HIP kernels.
    """

    @staticmethod
    def forward(ctx, x, gamma, beta):
        dims = (0, 2, 3)
        # torch's CPU BatchNorm (the reference's substrate) accumulates the batch
        # statistics in double (acc_type<float> on CPU) and rounds once to fp32.
        xd = x.double()
        mean_d = xd.mean(dim=dims)
        var_d = ((xd - mean_d[None, :, None, None]) ** 2).mean(dim=dims)
        mean = mean_d.to(x.dtype)
        var_b = var_d.to(x.dtype)
        invstd = torch.rsqrt(var_d + BN_EPS).to(x.dtype)
        xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
        ctx.save_for_backward(xhat, gamma, invstd)
        ctx.mark_non_differentiable(mean, var_b)
        return xhat * gamma[None, :, None, None] + beta[None, :, None, None], mean, var_b

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        xhat, gamma, invstd = ctx.saved_tensors
        dims = (0, 2, 3)
        dbeta = dy.double().sum(dim=dims).to(dy.dtype)
        dgamma = (dy.double() * xhat.double()).sum(dim=dims).to(dy.dtype)
        n = dy.numel() // dy.shape[1]
        dx = (gamma * invstd)[None, :, None, None] * (
            dy - (dbeta / n)[None, :, None, None] - xhat * (dgamma / n)[None, :, None, None])
        return dx, dgamma, dbeta


def _bn(x, st, prefix, training, ema_repeats, buffer_updates, tape, tag):
    gamma, beta = st[f"{prefix}.weight"], st[f"{prefix}.bias"]
    if training:
        out, mean, var_b = _BatchNormTrain.apply(x, gamma, beta)
        if buffer_updates is not None:
            with torch.no_grad():
                n = x.numel() // x.shape[1]
                rm = st[f"{prefix}.running_mean"].clone()
                rv = st[f"{prefix}.running_var"].clone()
                nbt = st[f"{prefix}.num_batches_tracked"].clone()
                var_u = var_b * (n / max(n - 1, 1))
                for _ in range(ema_repeats):
                    rm = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                    rv = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u
                    nbt = nbt + 1
                buffer_updates[f"{prefix}.running_mean"] = rm
                buffer_updates[f"{prefix}.running_var"] = rv
                buffer_updates[f"{prefix}.num_batches_tracked"] = nbt
        if tape is not None:
            tape[f"{tag}.mean"] = mean.detach()
            tape[f"{tag}.var"] = var_b.detach()
        return out
    mean = st[f"{prefix}.running_mean"]
    var_b = st[f"{prefix}.running_var"]
    inv = torch.rsqrt(var_b + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] \
        + beta[None, :, None, None]


# --------------------------------------------------------------------------
# bf16-operand arithmetic (what the HIP library's builder-chosen bf16 compute mode does; NOT the reference's arithmetic:
# its CPU path is float32 and on a GPU it autocasts to float16 with a GradScaler, train_model.py:131,144): every contraction (conv / convT, forward, input gradient, weight
# gradient) sees its two operands rounded to bfloat16 and accumulates in float32; everything else is float32.
# --------------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _ConvBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding, round_dx=False):
        ctx.save_for_backward(x, w)
        ctx.padding = padding
        ctx.round_dx = bool(round_dx)
        y = F.conv2d(_bf(x), _bf(w), b, padding=padding)
        return _bf(y) if _BF16_ROUND_OUTPUTS else y       # (straight-through: the gradient passes unchanged)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = _bf(dy)
        dx = torch.nn.grad.conv2d_input(x.shape, _bf(w), dyb, padding=ctx.padding)
        if ctx.round_dx:                                   # the input gradient is stored as a bfloat16 tensor
            dx = _bf(dx)
        dw = torch.nn.grad.conv2d_weight(_bf(x), w.shape, dyb, padding=ctx.padding)
        return dx, dw, dy.sum(dim=(0, 2, 3)), None, None


class _ConvT2x2BF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, round_dx=False):
        ctx.save_for_backward(x, w)
        ctx.round_dx = bool(round_dx)
        return F.conv_transpose2d(_bf(x), _bf(w), b, stride=2)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = _bf(dy)
        dx = F.conv2d(dyb, _bf(w), None, stride=2)                               # adjoint of the scatter
        if ctx.round_dx:                                   # the input gradient is stored as a bfloat16 tensor
            dx = _bf(dx)
        dw = torch.nn.grad.conv2d_weight(dyb, w.shape, _bf(x), stride=2)         # [cin][cout][2][2]
        return dx, dw, dy.sum(dim=(0, 2, 3)), None


_BF16_OPERANDS = False
_BF16_ROUND_OUTPUTS = False
_BF16_ROUND_GRADS = False
_WIDTHS_16 = False          # set by forward(): the model's first width is a multiple of 16 (then every width is)
_WIDTHS_32 = False          # ... of 32


class bf16_operands:
    """``with unet_ref.bf16_operands(): ...`` -- forward/backward of the oracle in the bf16-operand arithmetic.
    ``round_outputs=True``: every 3x3 conv output (conv + bias, before BatchNorm) is additionally rounded to bfloat16,
    as the HIP library's bf16 data flow does, which stores them as bf16; BatchNorm statistics are then those of the rounded values.
    ``round_grads`` (default: as ``round_outputs``; needs widths that are multiples of 16, as the library does): the
    input gradients of the 3x3 convs that feed a BatchNorm backward or the max-pool backward -- every second conv of
    a DoubleConv, and the first conv of the bottleneck and of encoders 2.. -- are rounded to bfloat16 as well (the library stores them as bf16), and so are the gradients that reach a
    DoubleConv's output from the max-pool backward + skip (encoders) and from a one-channel 1x1 head.  The decoder's
    first conv, whose input gradient feeds the transposed conv and the skip, keeps float32."""

    def __init__(self, round_outputs=False, round_grads=None):
        self.round_outputs = bool(round_outputs)
        self.round_grads = self.round_outputs if round_grads is None else bool(round_grads)

    def __enter__(self):
        global _BF16_OPERANDS, _BF16_ROUND_OUTPUTS, _BF16_ROUND_GRADS
        self.prev = (_BF16_OPERANDS, _BF16_ROUND_OUTPUTS, _BF16_ROUND_GRADS)
        _BF16_OPERANDS, _BF16_ROUND_OUTPUTS, _BF16_ROUND_GRADS = True, self.round_outputs, self.round_grads

    def __exit__(self, *exc):
        global _BF16_OPERANDS, _BF16_ROUND_OUTPUTS, _BF16_ROUND_GRADS
        _BF16_OPERANDS, _BF16_ROUND_OUTPUTS, _BF16_ROUND_GRADS = self.prev


def _conv3x3(x, w, b, round_dx=False):
    if _BF16_OPERANDS:
        return _ConvBF16.apply(x, w, b, 1, round_dx and _BF16_ROUND_GRADS and _WIDTHS_16)
    return F.conv2d(x, w, b, padding=1)


def _convt2x2(x, w, b):
    # (widths in whole 32-channel blocks: the library runs the transposed convs on its plane kernels -- their input gradient
    # and the gradient of the decoder's [up | skip] input are then bfloat16 tensors)
    return _ConvT2x2BF16.apply(x, w, b, _BF16_ROUND_GRADS and _WIDTHS_32) if _BF16_OPERANDS else F.conv_transpose2d(x, w, b, stride=2)


def _double_conv(x, st, prefix, training, ema_repeats, buffer_updates, tape, negative_slope=0.0):
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        x = _conv3x3(x, st[f"{prefix}.{conv_idx}.weight"], st[f"{prefix}.{conv_idx}.bias"],
                     round_dx=conv_idx == 3 or not prefix.startswith("decoder") or _WIDTHS_32)
        if tape is not None:
            tape[f"{prefix}.{conv_idx}.out"] = x
        x = _bn(x, st, f"{prefix}.{bn_idx}", training, ema_repeats, buffer_updates, tape,
                f"{prefix}.{bn_idx}")
        x = F.leaky_relu(x, negative_slope) if negative_slope else torch.relu(x)
        if tape is not None:
            tape[f"{prefix}.{bn_idx}.act"] = x
    return x


def forward(state, x_nchw, training=False, buffer_updates=None, tape=None, negative_slope=0.0,
            head_sigmoid=False):
    """Logits (N,out,H,W).  ``buffer_updates`` (dict) receives new BN buffers in train mode.
    ``negative_slope`` > 0: UNetDifferentActivation with LeakyReLU (models/unet.py:198-268);
    ``head_sigmoid``: UNetOverfit, which returns sigmoid(final_conv(.)) (models/unet.py:196)."""
    global _WIDTHS_16, _WIDTHS_32
    _, _, _, depth = infer_config(state)
    _WIDTHS_16 = state["encoder1.conv.conv.0.weight"].shape[0] % 16 == 0
    _WIDTHS_32 = state["encoder1.conv.conv.0.weight"].shape[0] % 32 == 0
    skips = []
    h = x_nchw
    ns = negative_slope
    for lvl in range(1, depth + 1):
        a = _double_conv(h, state, f"encoder{lvl}.conv.conv", training, 2, buffer_updates, tape, ns)
        if _BF16_OPERANDS and _BF16_ROUND_GRADS and _WIDTHS_16 and a.requires_grad:
            a.register_hook(_bf)          # the merged gradient (max-pool backward + skip) is stored as bfloat16
        skips.append(a)
        h = F.max_pool2d(a, kernel_size=2, stride=2)
    h = _double_conv(h, state, "bottleneck.conv", training, 1, buffer_updates, tape, ns)
    for lvl in range(depth, 0, -1):
        up = _convt2x2(h, state[f"decoder{lvl}.up.weight"], state[f"decoder{lvl}.up.bias"])
        if tape is not None:
            tape[f"decoder{lvl}.up.out"] = up
        h = torch.cat([up, skips[lvl - 1]], dim=1)
        h = _double_conv(h, state, f"decoder{lvl}.conv.conv", training, 1, buffer_updates, tape, ns)
    if _BF16_OPERANDS and _BF16_ROUND_GRADS and _WIDTHS_16 and h.requires_grad and state["final_conv.weight"].shape[0] == 1:
        h.register_hook(_bf)              # ... and the gradient the 1x1 head sends into the last DoubleConv
    out = F.conv2d(h, state["final_conv.weight"], state["final_conv.bias"])
    return torch.sigmoid(out) if head_sigmoid else out


def variant_forward(negative_slope=0.0, head_sigmoid=False):
    """``forward_fn`` for loss_and_grads / train_step of a U-Net variant."""
    def fn(state, x_nchw, training=False, buffer_updates=None, tape=None):
        return forward(state, x_nchw, training, buffer_updates, tape, negative_slope, head_sigmoid)
    return fn


# --------------------------------------------------------------------------
# loss + step
# --------------------------------------------------------------------------
def segmentation_loss(logits, target, smooth=1.0):
    """mean BCE-with-logits + dice over the whole flattened batch (train_model.py:120-128)."""
    x = logits.reshape(-1)
    y = target.reshape(-1).to(x.dtype)
    bce = (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()
    p = torch.sigmoid(x)
    inter = (p * y).sum()
    dice = 1 - (2.0 * inter + smooth) / (p.sum() + y.sum() + smooth)
    return bce + dice


def new_adam_state(state):
    return {"step": 0,
            "m": {k: torch.zeros_like(state[k]) for k in param_names(state)},
            "v": {k: torch.zeros_like(state[k]) for k in param_names(state)}}


def focal_loss(logits, target, alpha=0.25, gamma=2.0):
    """Sigmoid focal loss, mean over elements.  NOT in the reference: builder-defined as
    Lin et al. 2017 / the published torchvision.ops.sigmoid_focal_loss formula."""
    x = logits.reshape(-1)
    t = target.reshape(-1).to(x.dtype)
    p = torch.sigmoid(x)
    ce = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-x.abs()))
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.mean()


def loss_and_grads(state, x_nchw, y, training=True, tape=None, forward_fn=None, loss_fn=None):
    """``forward_fn`` defaults to the U-Net ``forward``; oracle/cnn_ref.py passes its own.  ``loss_fn``
    defaults to the reference's BCE + dice."""
    names = param_names(state)
    work = OrderedDict(state)
    leaves = []
    for k in names:
        t = state[k].detach().clone().requires_grad_(True)
        work[k] = t
        leaves.append(t)
    bufs = {}
    logits = (forward_fn or forward)(work, x_nchw, training=training, buffer_updates=bufs, tape=tape)
    loss = (loss_fn or segmentation_loss)(logits, y)
    grads = torch.autograd.grad(loss, leaves)
    return loss.detach(), logits.detach(), OrderedDict(zip(names, grads)), bufs


def clip_coefficient(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (||g||_2 + 1e-6), max=1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, coef


def train_step(state, adam, x_nchw, y, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=1e-5, clip=1.0, tape=None, forward_fn=None, loss_fn=None):
    """One optimisation step in place on ``state``/``adam``.  Returns dict of scalars+grads."""
    loss, logits, grads, bufs = loss_and_grads(state, x_nchw, y, training=True, tape=tape, forward_fn=forward_fn,
                                               loss_fn=loss_fn)
    total, coef = clip_coefficient(grads, clip)
    adam["step"] += 1
    t = adam["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    with torch.no_grad():
        for k, g in grads.items():
            g = g * coef
            p = state[k]
            # torch.optim.Adam's single-tensor CPU sequence (the one train_model.py:130,150 runs):
            # add(alpha=wd) -> lerp_ -> mul_/addcmul_ -> sqrt/div/add_ -> addcdiv_
            g = g.add(p, alpha=weight_decay)
            m = adam["m"][k].lerp_(g, 1 - b1)
            v = adam["v"][k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            state[k] = p.addcdiv(m, denom, value=-(lr / bc1))
        for k, v in bufs.items():
            state[k] = v
    return {"loss": float(loss), "grad_norm": float(total), "clip_coef": float(coef),
            "logits": logits, "grads": grads}


# --------------------------------------------------------------------------
# layout helpers shared by tests
# --------------------------------------------------------------------------
def nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def predict_mask(logits):
    """sigmoid > 0.5 (rfi_toolbox/scripts/evaluate_model.py:44-47) == logits > 0."""
    return (torch.sigmoid(logits) > 0.5)
