#!/usr/bin/env python3
"""One conv layer in a loop (GPU box): timing, or a target for `rocprofv3 --pmc` passes.
usage: bench_conv.py N H W CIN COUT [fwd|wgrad] [reps] [impl: 0 auto, 2 f32 MFMA, 3 bf16, 4 3xbf16]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from rfi_toolbox_amd._lib import check, lib  # noqa: E402
from rfi_toolbox_amd.runtime import Context  # noqa: E402

n, h, w, cin, cout = [int(v) for v in sys.argv[1:6]]
what = sys.argv[6] if len(sys.argv) > 6 else "fwd"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
impl = int(sys.argv[8]) if len(sys.argv) > 8 else 0
ctx = Context.get(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, h, w, cin), dtype=np.float32))
wt = ctx.to_device(rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * 0.05)
b = ctx.to_device(np.zeros(cout, np.float32))
sc = ctx.to_device(np.ones(cin, np.float32))
sh = ctx.to_device(np.zeros(cin, np.float32))
y = ctx.empty((n, h, w, cout), np.float32)
dy = ctx.to_device(rng.standard_normal((n, h, w, cout), dtype=np.float32))
gw = ctx.empty((cout, cin, 3, 3), np.float32)
P = lambda a: C.c_void_p(a.ptr)  # noqa: E731


def run():
    if what == "fwd":
        check(lib.rfi_op_conv3x3(ctx.handle, impl, P(x), n, h, w, cin, P(wt), P(b), cout, P(sc), P(sh), 1, P(y)))
    else:
        check(lib.rfi_op_conv3x3_wgrad(ctx.handle, impl, P(x), P(dy), n, h, w, cin, cout, P(sc), P(sh), 1, P(gw)))


for _ in range(3):
    run()
ctx.synchronize()
ctx.profile_reset()
ctx.profile(True)
for _ in range(reps):
    run()
ctx.synchronize()
ctx.profile(False)
rep = ctx.profile_report()
for k, v in rep.items():
    if v["flops"]:
        print(k, f"{v['ms'] / v['launches'] * 1e3:.1f} us/launch", f"{v['flops'] / v['ms'] / 1e9:.1f} TF")
