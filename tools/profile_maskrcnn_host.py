#!/usr/bin/env python3
"""Where the host time of a Mask R-CNN training step goes (GPU box): cProfile of a few steps + the per-launch profile of
the device kernels sorted by time.    python tools/profile_maskrcnn_host.py [batch]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from rfi_toolbox_amd.models import MaskRCNN  # noqa: E402
from rfi_toolbox_amd.runtime import Context  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = Context.get(0)
det = MaskRCNN(2, 3, 64, 256, 1024, seed=0)
x, targets = bench.synthetic_instances(B, 128, 0)
for _ in range(2):
    det.train_step(x, targets)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    det.train_step(x, targets)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
ctx.set_overlap(False)
ctx.profile_reset(); ctx.profile(True)
det.train_step(x, targets)
ctx.synchronize(); ctx.profile(False)
ctx.profile_dump("/tmp/mrcnn_launches.csv")
rows = bench.read_launch_csv("/tmp/mrcnn_launches.csv")
agg = {}
for r in rows:
    k = (r["family"], r["label"].split(" N")[0][:60])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += r["ms"]; a[2] += r["gflop"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{a[1]:8.3f} ms  {a[0]:4d} launches  {a[2] / max(a[1], 1e-9):8.1f} TF  {k[0]:18s} {k[1]}")
