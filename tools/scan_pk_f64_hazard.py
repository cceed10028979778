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

WINDOW = B.HAZARD_WINDOW          # instructions after the packed op (16 s_nop between the pair hid the fault: DESIGN.md)
count_pairs = B.count_pk_f64_pairs    # (the build itself runs this check on every device source it compiles)


def scan_source(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.HIPCC, *B.COMMON, *B.flags_for(src), "-x", "hip", "-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return count_pairs(B.isa_lines(out))


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
