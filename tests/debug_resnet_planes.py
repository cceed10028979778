"""Not a test (no test_ prefix): per-tensor gradient errors of the ResNet-encoder U-Net's bfloat16 flow against the oracle in
the same arithmetic, next to the float32-tensor bf16 mode.  python tests/debug_resnet_planes.py [f n s]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import resnet_unet_ref as rref          # noqa: E402
from oracle import unet_ref                         # noqa: E402
from rfi_toolbox_amd.models import UNetResNet18     # noqa: E402
from tests.test_gpu_resnet_unet import _inputs, _perturbed_state   # noqa: E402

f, n, s = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 4, 64)
st = _perturbed_state(f, 31)
x, y, xo, yo = _inputs(n, s, 32)
l32, lg32, g32, _ = rref.loss_and_grads(st, xo, yo)
with unet_ref.bf16_operands(round_outputs=True):
    lb, lgb, gb, _ = rref.loss_and_grads(st, xo, yo)
with unet_ref.bf16_operands():
    lr, lgr, gr, _ = rref.loss_and_grads(st, xo, yo)
m = UNetResNet18(3, 1, f).load_state_dict(st).train().set_compute_dtype("bfloat16")
m2 = UNetResNet18(3, 1, f).load_state_dict(st).train().set_compute_dtype("bfloat16_regs")
print("loss", m.forward_backward(x, y), float(lb), "| regs", m2.forward_backward(x, y), float(lr))
for k, g in gb.items():
    g = g.numpy().ravel()
    nrm = np.linalg.norm(g) + 1e-30
    a = np.linalg.norm(m.grad(k).ravel() - g) / nrm
    b = np.linalg.norm(g32[k].numpy().ravel() - g) / nrm
    gg = gr[k].numpy().ravel()
    c = np.linalg.norm(m2.grad(k).ravel() - gg) / (np.linalg.norm(gg) + 1e-30)
    print(f"{k:40s} planes-vs-oracle16 {a:9.4f}   f32-vs-oracle16 {b:9.4f}   regs-vs-oracleregs {c:9.4f}")
