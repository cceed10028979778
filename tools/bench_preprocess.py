#!/usr/bin/env python3
"""Secondary measurement (GPU box): on-device Preprocessor hot loop, GB/s against the 29 B/pixel
algorithmic traffic of SURVEY 8d (complex128 in, 3 x float32 out + 1 B label), device-resident."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rfi_toolbox_amd._lib import C128, C64, DEVICE, check, lib  # noqa: E402
from rfi_toolbox_amd.runtime import Context  # noqa: E402

ctx = Context.get(0)
out = {}
for name, dtype, code, n, ps in (("c128_256x128", np.complex128, C128, 256, 128), ("c64_256x128", np.complex64, C64, 256, 128),
                                 ("c128_4x1024", np.complex128, C128, 4, 1024)):
    rng = np.random.default_rng(0)
    z = (rng.normal(size=(n, ps, ps)) + 1j * rng.normal(size=(n, ps, ps))).astype(dtype)
    d_in, d_out = ctx.to_device(z), ctx.empty((n, ps, ps, 3), np.float32)
    for _ in range(3):
        check(lib.rfi_preprocess_patches(ctx.handle, C.c_void_p(d_in.ptr), DEVICE, code, n, ps, ps, C.c_void_p(d_out.ptr), DEVICE))
    ctx.synchronize()
    reps = 20
    ctx.timer_start()
    for _ in range(reps):
        check(lib.rfi_preprocess_patches(ctx.handle, C.c_void_p(d_in.ptr), DEVICE, code, n, ps, ps, C.c_void_p(d_out.ptr), DEVICE))
    ms = ctx.timer_stop() / reps
    px = n * ps * ps
    alg = px * (z.itemsize + 12 + 1)
    out[name] = {"ms": round(ms, 4), "Mpixel_per_s": round(px / ms / 1e3, 1), "algorithmic_GBps": round(alg / ms / 1e6, 1)}

# end-to-end Preprocessor.create_dataset on one 1024x1024 complex128 waterfall (4 views -> 256 patches
# of 128x128; host arrays in, host TorchDataset out): gather form (views/tiling/keep/labels on the GPU)
# against the host-bookkeeping form; the reference's own pieces are in BASELINE.md section 2
from rfi_toolbox_amd.preprocessing import Preprocessor  # noqa: E402

rng = np.random.default_rng(1)
z = (rng.normal(size=(1, 1, 1024, 1024)) + 1j * rng.normal(size=(1, 1, 1024, 1024)))
fl = rng.random((1, 1, 1024, 1024)) < 0.001
for name, kw in (("create_dataset_1024_gather", {}), ("create_dataset_1024_hostbook", {"on_device_tiling": False})):
    ts = []
    for i in range(4):
        np.random.seed(0)
        t0 = time.perf_counter()
        ds = Preprocessor(z, flags=fl).create_dataset(patch_size=128, **kw)
        ts.append(time.perf_counter() - t0)
    out[name] = {"patches": len(ds), "ms": round(min(ts[1:]) * 1e3, 2), "ms_per_patch": round(min(ts[1:]) * 1e3 / len(ds), 4)}
print(json.dumps(out))
