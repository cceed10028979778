#!/usr/bin/env python3
"""Idle time per HIP queue from a `rocprofv3 --kernel-trace` CSV: busy / span / gaps, and which kernels the gaps precede.

    python tools/trace_gaps.py path/to/*_kernel_trace.csv [fraction of events to skip at the start: 0.33]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.33
by = collections.defaultdict(list)
for r in rows:
    by[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, l in by.items():
    l.sort()
    l = l[int(len(l) * skip):]
    busy = sum(e - s for s, e, _ in l)
    span = l[-1][1] - l[0][0]
    gaps = [(max(l[i + 1][0] - l[i][1], 0), l[i + 1][2]) for i in range(len(l) - 1)]
    print(f"queue {q}: {len(l)} kernels, busy {busy / 1e6:.2f} ms, span {span / 1e6:.2f} ms, idle {sum(g for g, _ in gaps) / 1e6:.2f} ms")
    c = collections.defaultdict(lambda: [0, 0])
    for g, name in gaps:
        c[name][0] += g
        c[name][1] += 1
    for name, (t, n) in sorted(c.items(), key=lambda kv: -kv[1][0])[:10]:
        if t > 0:
            short = name.replace("rfi::(anonymous namespace)::", "").replace("void ", "")[:70]
            print(f"    idle before {short:70s} x{n:4d}  avg {t / 1e3 / n:7.2f} us  total {t / 1e6:6.2f} ms")
