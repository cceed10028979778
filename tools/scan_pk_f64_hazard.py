#!/usr/bin/env python3
"""Scan the gfx950 ISA of every device source for the instruction pair that miscomputes on MI355X under co-running load
(ROCm 7.2): a packed-fp32 VALU op (v_pk_*_f32) whose result is converted by v_cvt_f64_f32 a few instructions later
(observed: lanes 48-63 of the consumer occasionally see a stale source; tools/race_probe.py).  The elementwise
sources are built with -fno-slp-vectorize so the compiler forms no such pairs; this script is the check.

    python tools/scan_pk_f64_hazard.py            # exit status 1 if a pair is found"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rfi_toolbox_amd import build as B

WINDOW = 8


def scan_source(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.HIPCC, *B.COMMON, *B.flags_for(src), "-x", "hip", "-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        lines = [l.strip() for l in open(out) if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
    return count_pairs(lines)


def count_pairs(lines):
    n = 0
    for i, l in enumerate(lines):
        m = re.match(r"v_pk_\w+_f32\s+v\[(\d+):(\d+)\]", l)
        if not m:
            continue
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
        for j in range(i + 1, min(i + 1 + WINDOW, len(lines))):
            c = re.match(r"v_cvt_f64_f32\w*\s+v\[\d+:\d+\],\s+v(\d+)", lines[j])
            if c and int(c.group(1)) in dst:
                n += 1
                break
            w = re.match(r"v_\w+\s+v\[?(\d+)(?::(\d+))?\]?", lines[j])      # result overwritten: stop tracking those registers
            if w:
                dst -= set(range(int(w.group(1)), int(w.group(2) or w.group(1)) + 1))
    return n


def scan_all(workers=8):
    from concurrent.futures import ThreadPoolExecutor
    srcs = [s for s in B.SOURCES if s.endswith(".hip")]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return dict(zip(srcs, ex.map(scan_source, srcs)))


if __name__ == "__main__":
    res = scan_all()
    for src, n in res.items():
        print(f"{src}: {n} packed-fp32 -> v_cvt_f64_f32 pairs within {WINDOW} instructions")
    sys.exit(1 if any(res.values()) else 0)
