#!/usr/bin/env python3
"""Inference throughput (GPU box): eval-mode forward + sigmoid threshold + confusion counts on device
(rfi_model_eval_batch, the evaluate_model.py:18-58 path), batch 64 x 128 x 128 x 3 resident in HBM."""
import ctypes as C
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rfi_toolbox_amd._lib import DEVICE, check, lib  # noqa: E402
from rfi_toolbox_amd.data_generation import make_training_patches_device  # noqa: E402
from rfi_toolbox_amd.models import UNet  # noqa: E402
from rfi_toolbox_amd.runtime import Context  # noqa: E402

ctx = Context.get(0)
out = {}
for dtype in ("float32", "float32_mfma", "bfloat16"):
    torch.manual_seed(0)
    m = UNet(3, 1, 32).eval().set_compute_dtype(dtype)
    x, y = make_training_patches_device(64, 128, seed=1)
    tp, fp, fn = C.c_int64(), C.c_int64(), C.c_int64()

    def run():
        check(lib.rfi_model_eval_batch(m._h, C.c_void_p(x.ptr), DEVICE, C.c_void_p(y.ptr), DEVICE, 64, 128, 128, 0.5,
                                       C.byref(tp), C.byref(fp), C.byref(fn)))
    for _ in range(3):
        run()
    ctx.timer_start()
    reps = 20
    for _ in range(reps):
        run()
    ms = ctx.timer_stop() / reps
    out[dtype] = {"ms_per_batch": round(ms, 3), "patches_per_s": round(64 / ms * 1e3, 1)}
print(json.dumps(out))
