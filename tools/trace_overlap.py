#!/usr/bin/env python3
"""How well the streams of a training step overlap, from a `rocprofv3 --kernel-trace` CSV.

    python tools/trace_overlap.py path/to/*_kernel_trace.csv [steps in the trace: 10]

Takes the middle third of the trace and reports, per step: wall time, time with a matrix-core kernel (conv / wgrad / gemm)
active, time with only memory-bound kernels active, time with nothing active, and the per-queue busy times."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
n = len(ev)
ev = ev[n // 3: 2 * n // 3]
t0, t1 = ev[0][0], max(e[1] for e in ev)
steps = nsteps / 3.0
MF = ("conv_ws", "conv_stem", "gemm_ws", "conv_igemm", "wgrad_", "pconv", "pwgrad", "conv_direct")
pts = []
for s, e, name, q in ev:
    mf = any(k in name for k in MF)
    pts.append((s, 1, mf))
    pts.append((e, -1, mf))
pts.sort()
act_m = act_o = 0
last = t0
t_mf = t_mem = t_idle = t_both = 0
for t, d, mf in pts:
    dt = t - last
    if act_m and act_o: t_both += dt
    elif act_m: t_mf += dt
    elif act_o: t_mem += dt
    else: t_idle += dt
    last = t
    if mf: act_m += d
    else: act_o += d
ms = lambda v: v / 1e6 / steps
print(f"per step over {steps:.1f} steps: wall {ms(t1 - t0):.3f} ms | matrix-core kernel alone {ms(t_mf):.3f}, matrix + memory kernels together "
      f"{ms(t_both):.3f}, memory-bound kernels alone {ms(t_mem):.3f}, nothing running {ms(t_idle):.3f}")
byq = collections.defaultdict(lambda: [0, 0, 0])
for s, e, name, q in ev:
    byq[q][0] += e - s
    byq[q][1] += 1
    if any(k in name for k in MF): byq[q][2] += e - s
for q, (b, c, m) in byq.items():
    print(f"  queue {q}: {c / steps:.0f} kernels/step, busy {ms(b):.3f} ms/step (matrix-core kernels {ms(m):.3f})")
# concurrency of matrix-core kernels: time with two of them active at once
pts2 = sorted([(s, 1) for s, e, nme, q in ev if any(k in nme for k in MF)] + [(e, -1) for s, e, nme, q in ev if any(k in nme for k in MF)])
a = 0; last = t0; t2 = 0
for t, d in pts2:
    if a >= 2: t2 += t - last
    last = t; a += d
print(f"  two matrix-core kernels active at once: {ms(t2):.3f} ms/step")
