#!/usr/bin/env python3
"""Repeat forward + backward of one U-Net on one batch and report every gradient tensor (and a few intermediate
buffers) that is not bit-identical from repetition to repetition.  The backward pass runs on two streams and its kernels
share CUs, so any data race or co-residency hazard shows up here as a flicker; this is how the packed-fp32 ->
v_cvt_f64_f32 hazard of DESIGN.md was found (lanes 48-63 of the BatchNorm-backward sums of pool_bwd_merge, next to a
weight-gradient kernel).

    python tools/race_probe.py [bfloat16|float32] [repetitions] [comm_emulate world]
    RFI_SIDE_BOUND=2 ...      # bound the main stream's run-ahead over the side stream as the float32 path does"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def probe(mode="bfloat16", reps=300, emulate=0, features=16, batch=4, size=64, model="unet"):
    import torch
    from rfi_toolbox_amd.models import SimpleCNN, UNet, UNetResNet18
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(batch, size, size, 3, generator=g)
    y = (torch.rand(batch, size, size, generator=g) > 0.7).to(torch.uint8)
    torch.manual_seed(23)
    m = {"unet": UNet, "resnet": UNetResNet18, "cnn3": SimpleCNN}[model](3, 1, features).set_compute_dtype(mode)
    names = [n for n in m.state_dict() if "running" not in n and "num_batches" not in n]
    ref, bad = None, {}
    ctx.comm_emulate(emulate)
    try:
        for r in range(reps):
            m.forward_backward(x, y)
            cur = {n: np.array(m.grad(n), copy=True) for n in names}
            for t in ("dconcat.1",) if model == "unet" else ():
                cur["#" + t] = m.debug_tensor(t)
            if ref is None:
                ref = cur
                continue
            for n in cur:
                if not np.array_equal(ref[n], cur[n]):
                    idx = np.flatnonzero(ref[n] != cur[n])
                    d = float(np.abs(ref[n].astype(np.float64) - cur[n]).max())
                    bad.setdefault(n, []).append((r, int(idx.size), d, idx[:6].tolist()))
    finally:
        ctx.comm_emulate(0)
    return bad


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    emul = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = probe(mode, reps, emul)
    print(f"mode {mode} reps {reps} emulate {emul}: {len(bad)} tensors differed")
    for n, v in bad.items():
        print(n, v[:4], "..." if len(v) > 4 else "")
    sys.exit(1 if bad else 0)
