#!/usr/bin/env python3
"""Layer-by-layer parity probe (GPU box): forward conv outputs and backward dY of every conv of the
HIP U-Net against the fp64 oracle, next to the fp32 oracle's own error.  Debug aid, not a test."""
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import unet_ref  # noqa: E402
from rfi_toolbox_amd.models import UNet  # noqa: E402

f, n, size = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 2, 64))]
torch.manual_seed(1234)
m = UNet(3, 1, f)
st = m.state_dict()
g = torch.Generator().manual_seed(5)
x = torch.randn(n, size, size, 3, generator=g)
y = (torch.rand(n, size, size, generator=g) > 0.8).to(torch.uint8)
y[:, :, 10:14] = 1
if len(sys.argv) > 4:                      # golden fixture: its own initial state and batch
    gz = np.load(sys.argv[4])
    st = OrderedDict((k[7:], torch.from_numpy(gz[k])) for k in gz.files if k.startswith("state0/"))
    m.load_state_dict(st)
    x, y = torch.from_numpy(gz["img"]), torch.from_numpy(gz["lab"])
xo, yo = unet_ref.nhwc_to_nchw(x), y.float().unsqueeze(1)


def run(dtype):
    s = OrderedDict((k, (v.to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in st.items())
    names = unet_ref.param_names(s)
    for k in names:
        s[k] = s[k].clone().requires_grad_(True)
    tape = {}
    logits = unet_ref.forward(s, xo.to(dtype), training=True, buffer_updates={}, tape=tape)
    loss = unet_ref.segmentation_loss(logits, yo.to(dtype))
    keys = [k for k in tape if k.endswith(".out")]
    grads = torch.autograd.grad(loss, [tape[k] for k in keys] + [s[k] for k in names])
    gout = dict(zip(keys, grads[:len(keys)]))
    gpar = dict(zip(names, grads[len(keys):]))
    return tape, gout, gpar, float(loss), logits


t64, go64, gp64, l64, lg64 = run(torch.float64)
t32, go32, gp32, l32, lg32 = run(torch.float32)
loss = m.forward_backward(x, y)
print(f"loss hip {loss:.8f} ref32 {l32:.8f} ref64 {l64:.8f}")


def nhwc_flat(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy().ravel()


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


D = 4
fw = []
for l in range(1, D + 1):
    fw += [(f"encoder{l}.conv.conv.0.out", f"encY1.{l}"), (f"encoder{l}.conv.conv.3.out", f"encY2.{l}")]
fw += [("bottleneck.conv.0.out", "bottY1"), ("bottleneck.conv.3.out", "bottY2")]
for l in range(D, 0, -1):
    fw += [(f"decoder{l}.conv.conv.0.out", f"decY1.{l}"), (f"decoder{l}.conv.conv.3.out", f"decY2.{l}")]
print("--- forward conv outputs: rel err vs fp64 (hip | fp32 oracle)")
for ok, hk in fw:
    w64 = nhwc_flat(t64[ok])
    print(f"{hk:10s} hip {rel(m.debug_tensor(hk).astype(np.float64), w64):.2e}  ref32 {rel(nhwc_flat(t32[ok]).astype(np.float64), w64):.2e}")
print("logits     hip", rel(m.debug_tensor("logits").astype(np.float64), nhwc_flat(lg64)), " ref32",
      rel(nhwc_flat(lg32).astype(np.float64), nhwc_flat(lg64)))
print("--- backward dY at conv outputs (encoder + bottleneck survive the buffer reuse)")
bw = [("bottleneck.conv.3.out", "gBottA"), ("bottleneck.conv.0.out", "gBottB")]
for l in range(D, 0, -1):
    bw += [(f"encoder{l}.conv.conv.3.out", f"gA.{l}"), (f"encoder{l}.conv.conv.0.out", f"gB.{l}")]
for ok, hk in bw:
    w64 = nhwc_flat(go64[ok])
    print(f"{hk:10s} hip {rel(m.debug_tensor(hk).astype(np.float64), w64):.2e}  ref32 {rel(nhwc_flat(go32[ok]).astype(np.float64), w64):.2e}  scale {np.abs(w64).max():.2e}")
print("--- dconcat (grad of decoder conv1 input = [dUp | dSkip])")
for l in range(1, D + 1):
    up64 = go64[f"decoder{l}.up.out"]
    C = up64.shape[1]
    got = m.debug_tensor(f"dconcat.{l}").reshape(-1, 2 * C)[:, :C].ravel().astype(np.float64)
    print(f"dUp.{l}      hip {rel(got, nhwc_flat(up64)):.2e}  ref32 {rel(nhwc_flat(go32[f'decoder{l}.up.out']).astype(np.float64), nhwc_flat(up64)):.2e}")
print("--- parameter gradients")
for k in gp64:
    w64 = gp64[k].numpy()
    e_h = np.abs(m.grad(k) - w64).max()
    e_r = np.abs(gp32[k].numpy() - w64).max()
    flag = "  <<<" if e_h > 4 * e_r + 1e-6 * np.abs(w64).max() + 1e-9 else ""
    print(f"{k:38s} hip {e_h:.2e} ref32 {e_r:.2e} scale {np.abs(w64).max():.2e}{flag}")
