#!/usr/bin/env python3
"""Training throughput of the Mask R-CNN mask branch: MaskHead(C=256, 4 conv layers) on R RoIs of
14 x 14 x C RoIAlign-ed features resident in HBM; one step = forward + BCE + backward (incl. the input gradient) +
clip + Adam.  python tools/bench_mask_head.py [--rois 512] [--dtype f32|bf16]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    a = ap.parse_args()
    import torch
    from rfi_toolbox_amd._lib import Hyper
    from rfi_toolbox_amd.models import MaskHead
    from rfi_toolbox_amd.runtime import Context
    ctx = Context.get(0)
    torch.manual_seed(0)
    m = MaskHead(a.channels, 1, 4).train().set_compute_dtype("float32" if a.dtype == "f32" else "bfloat16")
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((a.rois, 14, 14, a.channels)).astype(np.float32))
    y = ctx.to_device((rng.random((a.rois, 28, 28)) > 0.5).astype(np.uint8))
    hp = Hyper(1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0)
    for _ in range(5):
        m.train_step_async(x.ptr, y.ptr, a.rois, 14, 14, hp)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.train_step_async(x.ptr, y.ptr, a.rois, 14, 14, hp)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fwd, step = m.algorithmic_flops(a.rois, 14, 14)
    print(json.dumps({"metric": "mask-head training RoIs/s", "value": round(a.rois / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                      "rois": a.rois, "channels": a.channels, "dtype": a.dtype, "tflops_whole_step": round(step / dt / 1e12, 1),
                      "loss": m.last_loss()[0]}))


if __name__ == "__main__":
    main()
