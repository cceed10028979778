"""``UNet`` with the reference's object protocol, executed by HIP kernels on MI355X.

Mirrors how the reference's callers use ``rfi_toolbox.models.UNet`` (models/unet.py:41-77):

    model = UNet(in_channels=3, out_channels=1, init_features=32).to("cuda")   # train_model.py:111
    model.train(); logits = model(x)            # x (N,C,H,W) float32 -> (N,1,H,W)   :136,145
    model.eval();  logits = model(x)                                                  # :157
    sd = model.state_dict(); model.load_state_dict(sd)       # :179, evaluate_model.py:35

``state_dict`` keys, shapes, dtypes and tensor layouts are the reference's, so checkpoints move
both ways.  Under ``torch.manual_seed(s)`` the constructor draws the initial weights with the same
torch initialisers in the same order as the reference's ``__init__`` and therefore produces
bit-identical initial weights.

What cannot be mirrored is ``loss.backward()``: there is no autograd here.  The loop body of
scripts/train_model.py:139-154 is one call, ``model.train_step(data, mask, lr=..., ...)``.
There is no CPU fallback: constructing a model without an MI355X raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict

import numpy as np

from .. import _lib
from .._lib import DEVICE, HOST, Hyper, check, lib
from ..runtime import Context, as_pointer, is_torch, torch


# ------------------------------------------------------------------ host-side parameter table
def _double_conv_entries(prefix, cin, cout):
    out = []
    for conv_i, bn_i, ci in ((0, 1, cin), (3, 4, cout)):
        out += [(f"{prefix}.{conv_i}.weight", (cout, ci, 3, 3), "conv_w"),
                (f"{prefix}.{conv_i}.bias", (cout,), "conv_b"),
                (f"{prefix}.{bn_i}.weight", (cout,), "bn_g"),
                (f"{prefix}.{bn_i}.bias", (cout,), "bn_b"),
                (f"{prefix}.{bn_i}.running_mean", (cout,), "bn_rm"),
                (f"{prefix}.{bn_i}.running_var", (cout,), "bn_rv"),
                (f"{prefix}.{bn_i}.num_batches_tracked", (), "bn_nbt")]
    return out


def unet_entries(in_channels, out_channels, init_features, depth=4):
    """(name, shape, kind) in the reference's ``state_dict`` order (unet.py:41-58, :79-98)."""
    f, ent, cin = init_features, [], in_channels
    for lvl in range(1, depth + 1):
        cout = f << (lvl - 1)
        ent += _double_conv_entries(f"encoder{lvl}.conv.conv", cin, cout)
        cin = cout
    ent += _double_conv_entries("bottleneck.conv", cin, 2 * cin)
    cin *= 2
    for lvl in range(depth, 0, -1):
        cout = f << (lvl - 1)
        ent += [(f"decoder{lvl}.up.weight", (cin, cout, 2, 2), "conv_w"),
                (f"decoder{lvl}.up.bias", (cout,), "conv_b")]
        ent += _double_conv_entries(f"decoder{lvl}.conv.conv", cin, cout)
        cin = cout
    ent += [("final_conv.weight", (out_channels, f, 1, 1), "conv_w"), ("final_conv.bias", (out_channels,), "conv_b")]
    return ent


def default_init_state(in_channels, out_channels, init_features, depth=4, entries=None):
    """torch's default Conv2d/ConvTranspose2d/BatchNorm2d initialisation, drawn from the global
    torch RNG in module-construction order == the reference's ``UNet.__init__`` draw order."""
    if torch is None:
        raise RuntimeError("torch is required for the default initialisation")
    sd = OrderedDict()
    bound = 0.0
    if entries is None:
        entries = unet_entries(in_channels, out_channels, init_features, depth)
    for name, shape, kind in entries:
        if kind == "conv_w":
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            sd[name] = w
        elif kind == "conv_b":
            b = torch.empty(shape)
            torch.nn.init.uniform_(b, -bound, bound)
            sd[name] = b
        elif kind in ("bn_g", "bn_rv"):
            sd[name] = torch.ones(shape)
        elif kind in ("bn_b", "bn_rm"):
            sd[name] = torch.zeros(shape)
        else:
            sd[name] = torch.tensor(0, dtype=torch.long)
    return sd


# ------------------------------------------------------------------ the model
class HipSegmenter:
    """Everything the models share: device binding, state_dict, forward, the optimisation step.
    A subclass supplies ``_entries`` (reference-order parameter table), ``_init`` (initial state),
    ``_create(ctx)`` (the C-ABI constructor) and ``_first_key`` (to recognise wrapped checkpoints)."""

    _first_key = ""
    _out_scale = 1                   # output map = _out_scale x input map (2 for the mask head)

    def _create(self, ctx):          # -> c_void_p model handle
        raise NotImplementedError

    def _setup(self, device):
        self.training = True
        self._h = None
        self.ctx = None
        self._bind(device)

    # ---- device binding
    def _bind(self, device):
        ctx = Context.get(device)
        h = self._create(ctx)
        old_state = self.state_dict() if self._h is not None else self._init
        self._release()
        self.ctx, self._h = ctx, h
        self._check_table()
        self.load_state_dict(old_state)
        self._init = None
        check(lib.rfi_model_set_training(self._h, 1 if self.training else 0))

    def _release(self):
        if self._h is not None:
            lib.rfi_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _check_table(self):
        n = C.c_int()
        check(lib.rfi_model_entry_count(self._h, C.byref(n)))
        if n.value != len(self._entries):
            raise RuntimeError("library/host parameter tables disagree")
        name, ndim, dims = C.c_char_p(), C.c_int(), (C.c_int64 * 4)()
        for i, (nm, shape, _) in enumerate(self._entries):
            check(lib.rfi_model_entry_info(self._h, i, C.byref(name), C.byref(ndim), dims, None, None))
            got = tuple(dims[k] for k in range(ndim.value))
            if name.value.decode() != nm or got != tuple(shape):
                raise RuntimeError(f"parameter table mismatch at {i}: {name.value!r}{got} vs {nm}{shape}")

    def to(self, device):
        """``.to("cuda")`` / ``.to("cuda:1")`` / ``.to(torch.device)``; 'cpu' raises (no CPU path)."""
        from ..runtime import _parse_device
        if _parse_device(device) != self.ctx.device_index:
            self._bind(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}")

    def set_compute_dtype(self, dtype="float32"):
        """Arithmetic of the conv / convT / weight-gradient contractions (storage, BatchNorm, loss, Adam are
        float32 in every mode).  ``"float32"`` (default): float32 by splitting every operand into three
        bfloat16 pieces, six bf16 MFMAs per product block -- float32-level accuracy (same error against
        float64 as the native path) at 2.7x its matrix rate.  ``"float32_mfma"``: the native float32 MFMA
        (exact fmaf chain).  ``"bfloat16"``: activations stored in HBM as bf16, bf16 MFMA operands, float32
        accumulate, float32 BatchNorm / loss / optimiser -- a builder-chosen reduced-precision mode, NOT the reference's
        arithmetic (its CPU path, the parity target, is float32; on a GPU it autocasts to float16 with a GradScaler,
        train_model.py:131,144); every contraction runs on the plane kernels (LDS-DMA staged operands; the transposed convs
        too at widths in whole 32-channel blocks); ``UNetResNet18`` has this data flow at widths in whole 16-channel chunks
        and falls back to ``"bfloat16_regs"`` at others, as the models without a plane flow do.  ``"float32_planes"``: the default arithmetic on pre-split plane tensors (same kernels as
        bfloat16, three pieces per value).  ``"bfloat16_regs"``: round 1's bf16 mode (float32 storage, operands
        rounded in registers)."""
        code = {"float32": 2, "fp32": 2, "f32": 2, "float32_3xbf16": 2, "float32_mfma": 0, "f32mfma": 0,
                "bfloat16": 1, "bf16": 1, "float32_planes": 3, "f32planes": 3, "bfloat16_regs": 4, "bf16regs": 4
                }.get(str(dtype).replace("torch.", ""))
        if code is None:
            raise ValueError("compute dtype must be float32, float32_mfma, float32_planes, bfloat16 or bfloat16_regs, "
                             f"got {dtype!r}")
        check(lib.rfi_model_set_compute_dtype(self._h, code))
        self.compute_dtype = {2: "float32", 0: "float32_mfma", 1: "bfloat16", 3: "float32_planes", 4: "bfloat16_regs"}[code]
        return self

    def set_loss(self, kind="bce_dice", alpha=0.25, gamma=2.0):
        """``"bce_dice"``: the reference's BCE-with-logits + dice (train_model.py:120-128, default);
        ``"focal"``: sigmoid focal loss, mean over elements (not in the reference; Lin et al. 2017)."""
        code = {"bce_dice": 0, "focal": 1}.get(kind)
        if code is None:
            raise ValueError(f"loss must be 'bce_dice' or 'focal', got {kind!r}")
        check(lib.rfi_model_set_loss(self._h, code, float(alpha), float(gamma)))
        return self

    # ---- mode
    def train(self, mode=True):
        self.training = bool(mode)
        check(lib.rfi_model_set_training(self._h, 1 if self.training else 0))
        return self

    def eval(self):
        return self.train(False)

    # ---- state
    def state_dict(self):
        sd = OrderedDict()
        for name, shape, kind in self._entries:
            if kind == "bn_nbt":
                v = np.zeros((), dtype=np.int64)
            else:
                v = np.empty(shape, dtype=np.float32)
            check(lib.rfi_model_store_entry(self._h, name.encode(), v.ctypes.data_as(C.c_void_p), v.nbytes))
            sd[name] = torch.from_numpy(v) if torch is not None else v
        return sd

    def load_state_dict(self, state_dict, strict=True):
        # accept the wrapped checkpoint train_model.py:177-183 writes as well as a bare state_dict
        if "model_state_dict" in state_dict and self._first_key not in state_dict:
            state_dict = state_dict["model_state_dict"]
        names = [e[0] for e in self._entries]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in set(names)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: Missing key(s): {missing}; "
                               f"Unexpected key(s): {unexpected}")
        for name, shape, kind in self._entries:
            if name not in state_dict:
                continue
            v = state_dict[name]
            v = v.detach().cpu().numpy() if is_torch(v) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: copying a param with shape {tuple(v.shape)} "
                                   f"from checkpoint, the shape in current model is {tuple(shape)}")
            v = np.ascontiguousarray(v, dtype=np.int64 if kind == "bn_nbt" else np.float32)
            check(lib.rfi_model_load_entry(self._h, name.encode(), v.ctypes.data_as(C.c_void_p), v.nbytes))
        return self

    def named_parameters(self):
        sd = self.state_dict()
        for name, _, kind in self._entries:
            if kind in ("conv_w", "conv_b", "bn_g", "bn_b"):
                yield name, sd[name]

    def parameters(self):
        """Host snapshots of the parameters (copies: updates happen on the GPU in train_step)."""
        for _, p in self.named_parameters():
            yield p

    def num_parameters(self) -> int:
        n = C.c_int64()
        check(lib.rfi_model_param_count(self._h, C.byref(n)))
        return n.value

    # ---- forward
    def _forward(self, x, nchw):
        shape = tuple(x.shape)
        if len(shape) != 4:
            raise ValueError(f"expected a 4-D input, got shape {shape}")
        n, c, h, w = (shape if nchw else (shape[0], shape[3], shape[1], shape[2]))
        if c != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {c}")
        ptr, mem, keep = as_pointer(x, np.float32, self.ctx)
        ho, wo = h * self._out_scale, w * self._out_scale
        out = np.empty((n, self.out_channels, ho, wo) if nchw else (n, ho, wo, self.out_channels), dtype=np.float32)
        fn = lib.rfi_model_forward_nchw if nchw else lib.rfi_model_forward_nhwc
        check(fn(self._h, C.c_void_p(ptr), mem, n, h, w, out.ctypes.data_as(C.c_void_p), HOST))
        del keep
        if is_torch(x):
            t = torch.from_numpy(out)
            return t.to(x.device) if x.is_cuda else t
        return out

    def forward(self, x):
        """x (N,C,H,W) float32 -> logits (N,out,H,W)  (unet.py:60-77)."""
        return self._forward(x, nchw=True)

    __call__ = forward

    def forward_nhwc(self, x):
        """x (N,H,W,C) as ``Preprocessor`` emits it -> logits (N,H,W,out); no transposes anywhere."""
        return self._forward(x, nchw=False)

    # ---- the optimisation step of scripts/train_model.py:139-154
    @staticmethod
    def _hyper(lr, betas, eps, weight_decay, max_grad_norm):
        return Hyper(float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                     float(max_grad_norm))

    def _xy(self, data, mask, nhwc):
        shape = tuple(data.shape)
        if len(shape) != 4:
            raise ValueError(f"expected 4-D data, got shape {shape}")
        if not nhwc:
            raise ValueError("train_step takes NHWC data (N,H,W,C) as Preprocessor emits it")
        n, h, w, c = shape
        if c != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {c}")
        mshape = tuple(mask.shape)
        ho, wo = h * self._out_scale, w * self._out_scale
        if mshape not in ((n, ho, wo), (n, 1, ho, wo), (n, ho, wo, 1)):
            raise ValueError(f"mask shape {mshape} does not match data {shape}")
        xp, xm, k1 = as_pointer(data, np.float32, self.ctx)
        yp, ym, k2 = as_pointer(mask, np.uint8, self.ctx)
        return n, h, w, xp, xm, yp, ym, (k1, k2)

    def train_step(self, data, mask, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
                   max_grad_norm=1.0):
        """zero_grad, forward, BCEWithLogits+dice, backward, clip_grad_norm_, Adam step; returns the loss.
        Defaults are the reference script's (train_model.py:89,95,130,149).  data: NHWC float32,
        mask: (N,H,W) uint8/bool, non-zero == RFI."""
        n, h, w, xp, xm, yp, ym, keep = self._xy(data, mask, True)
        hp = self._hyper(lr, betas, eps, weight_decay, max_grad_norm)
        loss = C.c_float()
        check(lib.rfi_train_step(self._h, C.c_void_p(xp), xm, C.c_void_p(yp), ym, n, h, w, C.byref(hp),
                                 C.byref(loss)))
        del keep
        return loss.value

    def loss(self, data, mask):
        """BCE+dice of the current mode's forward, no update (validation loop, train_model.py:157-167)."""
        n, h, w, xp, xm, yp, ym, keep = self._xy(data, mask, True)
        loss = C.c_float()
        check(lib.rfi_model_loss(self._h, C.c_void_p(xp), xm, C.c_void_p(yp), ym, n, h, w, C.byref(loss)))
        del keep
        return loss.value

    def forward_backward(self, data, mask):
        n, h, w, xp, xm, yp, ym, keep = self._xy(data, mask, True)
        loss = C.c_float()
        check(lib.rfi_train_forward_backward(self._h, C.c_void_p(xp), xm, C.c_void_p(yp), ym, n, h, w,
                                             C.byref(loss)))
        del keep
        return loss.value

    def apply_gradients(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, max_grad_norm=1.0,
                        grad_scale=1.0):
        hp = self._hyper(lr, betas, eps, weight_decay, max_grad_norm)
        norm = C.c_float()
        check(lib.rfi_train_apply(self._h, C.byref(hp), float(grad_scale), C.byref(norm)))
        return norm.value

    def accumulate_gradients(self, phase):
        """Sum the gradients of several backward passes (each pass overwrites the gradient buffer): ``"begin"`` zeroes
        the accumulator, ``"add"`` after each backward pass, ``"end"`` makes the sum the gradient ``apply_gradients`` sees."""
        check(lib.rfi_model_grad_accumulate(self._h, {"begin": 0, "add": 1, "end": 2}[phase]))

    def allreduce_gradients(self):
        check(lib.rfi_model_allreduce_grads(self._h))

    def train_step_async(self, data_dev, mask_dev, n, h, w, hyper: Hyper):
        """Enqueue one full step on device-resident inputs without any host sync (bench loops)."""
        check(lib.rfi_train_step_async(self._h, C.c_void_p(data_dev), C.c_void_p(mask_dev), n, h, w,
                                       C.byref(hyper)))

    def last_loss(self):
        a, b = C.c_float(), C.c_float()
        check(lib.rfi_model_last_loss(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def grad(self, name) -> np.ndarray:
        """Gradient of parameter ``name`` after forward_backward, in the reference's layout."""
        shape = dict((e[0], e[1]) for e in self._entries)[name]
        g = np.empty(shape, dtype=np.float32)
        check(lib.rfi_model_store_grad(self._h, name.encode(), g.ctypes.data_as(C.c_void_p), g.nbytes))
        return g

    def adam_state(self, name):
        shape = dict((e[0], e[1]) for e in self._entries)[name]
        m, v, step = np.empty(shape, np.float32), np.empty(shape, np.float32), C.c_int64()
        check(lib.rfi_model_store_adam(self._h, name.encode(), m.ctypes.data_as(C.c_void_p),
                                       v.ctypes.data_as(C.c_void_p), m.nbytes, C.byref(step)))
        return m, v, step.value

    def debug_tensor(self, name) -> np.ndarray:
        """Flat copy of an internal activation/gradient buffer (see rfi_model_debug_tensor)."""
        n = C.c_int64()
        check(lib.rfi_model_debug_tensor(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        check(lib.rfi_model_debug_tensor(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.size,
                                         C.byref(n)))
        return out

    def load_adam_state(self, name, m, v):
        m = np.ascontiguousarray(m, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        check(lib.rfi_model_load_adam(self._h, name.encode(), m.ctypes.data_as(C.c_void_p),
                                      v.ctypes.data_as(C.c_void_p), m.nbytes))

    def set_adam_step(self, step):
        check(lib.rfi_model_set_adam_step(self._h, int(step)))

    def optimizer_state_dict(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        """torch.optim.Adam-format state (what train_model.py:180 stores): parameter i is the
        i-th entry of ``named_parameters()``, exactly torch's numbering of ``model.parameters()``."""
        names = [n for n, _, k in self._entries if k in ("conv_w", "conv_b", "bn_g", "bn_b")]
        state, step = {}, 0
        for i, n in enumerate(names):
            m, v, step = self.adam_state(n)
            if step:
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": torch.from_numpy(m),
                            "exp_avg_sq": torch.from_numpy(v)}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd):
        names = [n for n, _, k in self._entries if k in ("conv_w", "conv_b", "bn_g", "bn_b")]
        step = 0
        for i, st in osd.get("state", {}).items():
            n = names[int(i)]
            self.load_adam_state(n, st["exp_avg"].detach().cpu().numpy(), st["exp_avg_sq"].detach().cpu().numpy())
            step = max(step, int(float(st["step"])))
        self.set_adam_step(step)

    def eval_batch(self, data, mask, threshold=0.5):
        """(tp, fp, fn) of one batch: forward, sigmoid > threshold, counts -- all on the GPU."""
        n, h, w, xp, xm, yp, ym, keep = self._xy(data, mask, True)
        tp, fp, fn = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.rfi_model_eval_batch(self._h, C.c_void_p(xp), xm, C.c_void_p(yp), ym, n, h, w, float(threshold),
                                       C.byref(tp), C.byref(fp), C.byref(fn)))
        del keep
        return tp.value, fp.value, fn.value

    def algorithmic_flops(self, n, h, w):
        f, s = C.c_double(), C.c_double()
        check(lib.rfi_model_algorithmic_flops(self._h, n, h, w, C.byref(f), C.byref(s)))
        return f.value, s.value


class UNet(HipSegmenter):
    """MI355X-native U-Net; same constructor signature as the reference (unet.py:42)."""

    _DEPTH = 4
    _first_key = "encoder1.conv.conv.0.weight"

    def __init__(self, in_channels=1, out_channels=1, init_features=32, *, device=None, depth=None):
        for v, nm in ((in_channels, "in_channels"), (out_channels, "out_channels"), (init_features, "init_features")):
            if not isinstance(v, (int, np.integer)) or v <= 0:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        self.in_channels, self.out_channels, self.init_features = int(in_channels), int(out_channels), int(init_features)
        self.depth = int(depth if depth is not None else self._DEPTH)
        self._entries = unet_entries(self.in_channels, self.out_channels, self.init_features, self.depth)
        self._init = default_init_state(self.in_channels, self.out_channels, self.init_features, self.depth)
        self._setup(device)

    _NEGATIVE_SLOPE = 0.0      # activation after every BatchNorm: 0 = ReLU
    _HEAD_SIGMOID = False

    def _create(self, ctx):
        h = C.c_void_p()
        check(lib.rfi_unet_create(ctx.handle, self.in_channels, self.out_channels, self.init_features,
                                  self.depth, C.byref(h)))
        slope = float(getattr(self, "negative_slope", self._NEGATIVE_SLOPE))
        if slope:
            check(lib.rfi_model_set_activation(h, slope))
        if self._HEAD_SIGMOID:
            check(lib.rfi_model_set_head_sigmoid(h, 1))
        return h


class UNetBigger(UNet):
    """5-level variant (reference models/unet.py:79-118)."""
    _DEPTH = 5


class UNetOverfit(UNet):
    """Reference models/unet.py:156-196: five levels, ``init_features=128`` by default, and the forward
    returns ``sigmoid(final_conv(.))``.  ``train_step`` feeds that output to BCE-with-logits + dice, as
    scripts/train_model.py:120,146 does with whatever the model returns."""
    _DEPTH = 5
    _HEAD_SIGMOID = True

    def __init__(self, in_channels=1, out_channels=1, init_features=128, *, device=None):
        super().__init__(in_channels, out_channels, init_features, device=device)


def _slope_of(activation):
    """negative_slope of the activation the reference would build with ``activation(inplace=True)``."""
    if activation is None:
        return 0.0
    if isinstance(activation, (int, float)):
        return float(activation)
    import functools
    nn = torch.nn
    if isinstance(activation, functools.partial):
        if activation.func is nn.LeakyReLU:
            return float(activation.keywords.get("negative_slope", activation.args[0] if activation.args else 0.01))
        activation = activation.func
    if isinstance(activation, nn.Module):
        if isinstance(activation, nn.LeakyReLU):
            return float(activation.negative_slope)
        if isinstance(activation, nn.ReLU):
            return 0.0
    if activation is nn.ReLU:
        return 0.0
    if activation is nn.LeakyReLU:
        return 0.01
    raise ValueError(f"activation {activation!r} is not available on the device path "
                     "(ReLU and LeakyReLU are)")


class UNetDifferentActivation(UNet):
    """Reference models/unet.py:198-268: the U-Net with ``activation(inplace=True)`` after every
    BatchNorm.  ``activation``: ``torch.nn.ReLU`` (default), ``torch.nn.LeakyReLU`` (slope 0.01), a
    ``functools.partial(nn.LeakyReLU, negative_slope=s)``, an instance of either, or the slope itself."""

    def __init__(self, in_channels=1, out_channels=1, init_features=32, activation=None, *, device=None):
        self.negative_slope = _slope_of(activation)
        if not 0.0 <= self.negative_slope < 1.0:
            raise ValueError("negative_slope must be in [0, 1)")
        super().__init__(in_channels, out_channels, init_features, device=device)
