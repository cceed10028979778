#!/usr/bin/env python3
"""PCIe-inclusive rate of the training step (GPU box): `UNet.train_step(data, mask)` with HOST numpy buffers at the C-ABI
boundary (rfi_train_step with x_mem = host: upload of 64 x 128 x 128 x 3 float32 + the uint8 mask, the step, the loss read
back) next to the device-resident `rfi_train_step_async` loop bench.py times.  The bench line's `value` is the resident
figure; this number is the note DESIGN.md §5 carries beside it."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from rfi_toolbox_amd._lib import Hyper  # noqa: E402
from rfi_toolbox_amd.data_generation import make_training_patches_device  # noqa: E402
from rfi_toolbox_amd.models import UNet  # noqa: E402
from rfi_toolbox_amd.runtime import Context  # noqa: E402

ctx = Context.get(0)
B, S = 64, 128
out = {"batch": B, "patch": [S, S, 3]}
for dtype in ("float32", "bfloat16"):
    torch.manual_seed(0)
    m = UNet(3, 1, 32).train().set_compute_dtype(dtype)
    xd, yd = make_training_patches_device(B, S, seed=1)
    xh, yh = xd.numpy(), yd.numpy()                         # pageable host copies of the same batch
    xh = np.ascontiguousarray(xh, dtype=np.float32)
    yh = np.ascontiguousarray(yh, dtype=np.uint8)
    for _ in range(5):
        m.train_step(xh, yh)
    ctx.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        m.train_step(xh, yh)                                # synchronous: returns the loss
    ctx.synchronize()
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    hp = Hyper(1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0)
    for _ in range(5):
        m.train_step_async(xd.ptr, yd.ptr, B, S, S, hp)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.train_step_async(xd.ptr, yd.ptr, B, S, S, hp)
    ctx.synchronize()
    dev_ms = (time.perf_counter() - t0) / reps * 1e3
    out[dtype] = {"host_buffers_ms_per_step": round(host_ms, 3), "host_buffers_patches_per_s": round(B / host_ms * 1e3, 1),
                  "resident_ms_per_step": round(dev_ms, 3), "resident_patches_per_s": round(B / dev_ms * 1e3, 1),
                  "upload_bytes_per_step": int(xh.nbytes + yh.nbytes)}
print(json.dumps(out))
