#!/usr/bin/env python3
"""One training step as a timeline, from a `rocprofv3 --kernel-trace` CSV: every kernel of a middle step with its queue,
start (us from the step's first kernel), duration, the gap to the previous kernel of the same queue, and whether a
matrix-core kernel was running on another queue meanwhile.

    python tools/trace_timeline.py path/to/*_kernel_trace.csv [marker kernel: adam_kernel]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
marks = [i for i, e in enumerate(ev) if marker in e[2]]
mid = len(marks) // 2
step = ev[marks[mid - 1] + 1: marks[mid] + 1]
MF = ("conv_ws", "conv_stem", "gemm_ws", "conv_igemm", "wgrad_", "pconv", "pwgrad", "conv_direct")
t0 = step[0][0]
queues = sorted({e[3] for e in step})
last_end = {}
short = lambda n: re.sub(r"\(.*", "", re.sub(r"rfi::\(anonymous namespace\)::|void |rfi::", "", n))[:58]
print(f"step of {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us; queues {queues}")
for s, e, name, q in step:
    other_mf = any(o[3] != q and any(k in o[2] for k in MF) and o[0] < e and o[1] > s for o in step)
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"q{queues.index(q)} {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {'|mfma elsewhere' if other_mf else '               '}  {short(name)}")
