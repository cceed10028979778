#!/usr/bin/env python3
"""Print how far the HIP path is from the committed f=4 golden trajectory (GPU box; debug aid used
to calibrate the tolerances of tests/test_gpu_unet.py::test_three_steps_golden_f4)."""
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from rfi_toolbox_amd.models import UNet  # noqa: E402

g = np.load("tests/golden/unet_f4_b4_s32.npz")


def prebn(k):          # conv biases feeding a BatchNorm: exact gradient is 0, both sides hold noise
    return k.endswith(".bias") and (k.endswith("conv.0.bias") or k.endswith("conv.3.bias"))


st = OrderedDict((k[7:], torch.from_numpy(g[k])) for k in g.files if k.startswith("state0/"))
m = UNet(3, 1, 4).load_state_dict(st).train()
lr, b1, b2, eps, wd, clip = [float(v) for v in g["hyper"]]
print("relu margins", g["relu_margin"])
for s in (1, 2, 3):
    loss = m.forward_backward(g["img"], g["lab"])
    if s == 1:
        gn = float(g["grad_norms"][0]); coef = min(1.0, clip / (gn + 1e-6))
        rels = {k[6:]: float(np.linalg.norm(m.grad(k[6:]) * coef - g[k]) / (np.linalg.norm(g[k]) + 1e-30))
                for k in g.files if k.startswith("grad1/") and not prebn(k[6:])}
        big = sorted(rels.items(), key=lambda kv: -kv[1])[:6]
        print("grad rel-L2 worst:", big)
    norm = m.apply_gradients(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, max_grad_norm=clip)
    print(f"step {s}: loss {loss:.8f} want {g['losses'][s-1]:.8f}  norm {norm:.6f} want {g['grad_norms'][s-1]:.6f}")
    if s in (1, 3):
        worst, frac = 0, 0
        for k, v in m.state_dict().items():
            if k.endswith("num_batches_tracked") or prebn(k):
                continue
            d = np.abs(v.numpy() - g[f"state{s}/{k}"])
            if d.max() > worst:
                worst, wk = d.max(), k
            frac = max(frac, float((d > 2e-5).mean()))
        print(f"  state{s}: worst |d| {worst:.3e} ({wk}), worst fraction of elements > 2e-5: {frac:.4f}")
        m.eval(); ev = m.forward_nhwc(g["img"]); m.train()
        print(f"  eval logits max |d| {np.abs(ev[..., 0] - g[f'logits_eval{s}'][:, 0]).max():.3e}")
for k in ("encoder1.conv.conv.0.weight", "decoder2.up.weight", "final_conv.weight"):
    mm, vv, step = m.adam_state(k)
    wm, wv = g[f"adam_m3/{k}"], g[f"adam_v3/{k}"]
    print(k, "m rel", np.linalg.norm(mm - wm) / np.linalg.norm(wm), "v rel", np.linalg.norm(vv - wv) / np.linalg.norm(wv))
