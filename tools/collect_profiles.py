#!/usr/bin/env python3
"""Fold gpurun_out/<TAG>/ (written on the GPU box by tools/profile_round.sh) into the tracked profiles/<TAG>_* files.

    python tools/collect_profiles.py r2
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(src, "bench*.json"))):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
for d, name in (("stats_serial", "kernel_stats"), ("stats_overlap", "kernel_stats_overlap"),
                ("stats_serial_bf16", "kernel_stats_bf16"), ("stats_overlap_bf16", "kernel_stats_overlap_bf16"),
                ("stats_serial_cnn3", "kernel_stats_cnn3"), ("stats_serial_unet1024", "kernel_stats_unet1024"),
                ("stats_serial_resnet1024", "kernel_stats_resnet1024"), ("stats_serial_maskrcnn", "kernel_stats_maskrcnn")):
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:                       # (gpurun merges every call's files into the same tree: the newest run's table)
        shutil.copy(max(hits, key=os.path.getmtime), os.path.join(dst, f"{tag}_{name}.csv"))
here = os.path.join(ROOT, "tools")
for dt, suffix in (("f32", ""), ("bf16", "_bf16"), ("cnn3", "_cnn3"), ("unet1024", "_unet1024_bf16"), ("resnet1024", "_resnet1024_bf16"),
                   ("maskrcnn", "_maskrcnn")):
    fd, wd, md = (os.path.join(src, f"pmc_{k}_{dt}") for k in ("fetch", "write", "mfma"))
    if os.path.isdir(fd) and os.path.isdir(wd):
        subprocess.check_call([sys.executable, os.path.join(here, "summarize_pmc.py"), fd, wd,
                               os.path.join(dst, f"{tag}_traffic{suffix}.json")], stdout=subprocess.DEVNULL)
    if os.path.isdir(md):
        subprocess.check_call([sys.executable, os.path.join(here, "summarize_mfma.py"), md,
                               os.path.join(dst, f"{tag}_mfma_util{suffix}.json")], stdout=subprocess.DEVNULL)
    # summaries written on the GPU box by tools/profile_round.sh (the raw counter tables do not travel back)
    for kind, name in (("traffic", "traffic"), ("mfma_util", "mfma_util")):
        f = os.path.join(src, f"{kind}_{dt}.json")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{tag}_{name}{suffix}.json"))
print("\n".join(sorted(os.listdir(dst))))
