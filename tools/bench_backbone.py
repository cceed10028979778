#!/usr/bin/env python3
"""Forward + backward throughput of the ResNet-50-FPN backbone (BASELINE configs[3] shape): batch 64 x
128 x 128 x 3 patches resident in HBM, gradients of all five pyramid levels given.  python tools/bench_backbone.py"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    a = ap.parse_args()
    import torch
    from rfi_toolbox_amd._lib import DEVICE, check, lib
    from rfi_toolbox_amd.models import ResNet50FPN
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    torch.manual_seed(0)
    m = ResNet50FPN(3, 64, 256).set_compute_dtype("float32" if a.dtype == "f32" else "bfloat16")
    rng = np.random.default_rng(0)
    n, s = a.batch, a.size
    x = ctx.to_device(rng.standard_normal((n, s, s, 3)).astype(np.float32))
    shapes = [(n, s >> (2 + i), s >> (2 + i), 256) for i in range(5)]
    outs = [ctx.empty(sh, np.float32) for sh in shapes]
    douts = [ctx.to_device((rng.standard_normal(sh) * 1e-3).astype(np.float32)) for sh in shapes]
    po = (C.c_void_p * 5)(*[o.ptr for o in outs])
    pd = (C.c_void_p * 5)(*[d.ptr for d in douts])

    def step():
        check(lib.rfi_backbone_forward(m._h, C.c_void_p(x.ptr), DEVICE, n, s, s, po, DEVICE))
        check(lib.rfi_backbone_backward(m._h, C.c_void_p(x.ptr), DEVICE, n, s, s, pd, DEVICE))
    for _ in range(3):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fwd, st = m.algorithmic_flops(n, s, s)
    print(json.dumps({"metric": "ResNet-50-FPN forward+backward patches/s", "value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                      "batch": n, "size": s, "dtype": a.dtype, "gflop_fwd_per_patch": round(fwd / n / 1e9, 2),
                      "tflops_whole_step": round(st / dt / 1e12, 1)}))


if __name__ == "__main__":
    main()
