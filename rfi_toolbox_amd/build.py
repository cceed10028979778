"""Build librfi_hip.so (gfx950) in-tree with hipcc.

    python -m rfi_toolbox_amd.build            # incremental
    python -m rfi_toolbox_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  Objects go to build/obj, the shared library to
rfi_toolbox_amd/librfi_hip.so (git-ignored; it travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "librfi_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["api.cpp", "model.cpp", "model_cnn.cpp", "model_planes.cpp", "model_resnet.cpp", "model_mask.cpp", "model_backbone.cpp", "model_mlp.cpp", "elem_kernels.hip", "conv_direct.hip", "conv_mfma.hip",
           "wgrad_mfma.hip", "preprocess.hip", "synth.hip", "order_stats.hip", "planes_elem.hip", "conv_planes.hip", "conv_ws.hip", "conv_stem.hip", "gemm_ws.hip", "wgrad_ws.hip", "wgrad_stem.hip", "wgrad_planes.hip", "wgrad_split.hip", "detect_kernels.hip", "detect_sample.hip", "rpn_kernels.hip", "resnet_kernels.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "model.hpp", "planes.hpp", "ws_common.hpp", os.path.join(ROOT, "include", "rfi_hip.h")]

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-ffp-contract=off"]
if os.environ.get("RFI_DIAG_STAMPS"):        # diagnostic build with in-kernel cycle stamps (never shipped)
    COMMON.append("-DRFI_DIAG_STAMPS=1")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# ROCm 7.2 / gfx950: a packed-fp32 VALU result (v_pk_mul_f32 ...) converted by v_cvt_f64_f32 a few instructions later is
# occasionally read stale in lanes 48-63 when another kernel shares the CU (found with tools/race_probe.py: the fp64
# BatchNorm-backward sums of pool_bwd_merge next to a weight-gradient kernel).  The sources below accumulate
# float32 products in fp64, so they are built without the SLP vectoriser (which forms the packed ops); they are bound
# by HBM, the packed forms bought nothing.  tools/scan_pk_f64_hazard.py checks the ISA of every source for the pair.
NO_SLP = {"elem_kernels.hip", "planes_elem.hip", "resnet_kernels.hip", "rpn_kernels.hip", "detect_kernels.hip",
          "preprocess.hip", "order_stats.hip", "synth.hip", "conv_direct.hip",
          # every other source that converts float32 to double (the statistics records of the conv epilogues, slab sums)
          "conv_planes.hip", "wgrad_planes.hip", "wgrad_split.hip", "wgrad_mfma.hip", "conv_mfma.hip",
          # conv_ws.hip: its producer waves run VALU next to the consumer waves' MFMAs, where a packed fp32 op costs
          # ~13 cycles more than the two scalar ops it replaces (MI355X guide, "price of one filler beside MFMAs")
          "conv_ws.hip", "gemm_ws.hip", "wgrad_ws.hip", "conv_stem.hip", "wgrad_stem.hip"}


def flags_for(src):
    return ["-fno-slp-vectorize"] if src in NO_SLP else []


# ---- ISA check run on every device source as it is compiled (tools/scan_pk_f64_hazard.py is the command-line form): the
# gfx950 instruction pair that miscomputes under co-running load -- a packed-fp32 VALU result (or a v_mov copy of it)
# converted by v_cvt_f64_f32 within HAZARD_WINDOW instructions.  A source that contains one fails the build.
HAZARD_WINDOW = 20


def count_pk_f64_pairs(lines, window=HAZARD_WINDOW):
    import re
    n = 0
    for i, l in enumerate(lines):
        m = re.match(r"v_pk_\w+_f32\s+v\[(\d+):(\d+)\]", l)
        if not m:
            continue
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))      # registers holding the packed result (or a copy of it)
        for j in range(i + 1, min(i + 1 + window, len(lines))):
            if not dst:
                break
            c = re.match(r"v_cvt_f64_f32\w*\s+v\[\d+:\d+\],\s+v(\d+)", lines[j])
            if c and int(c.group(1)) in dst:
                n += 1
                break
            mv = re.match(r"v_mov_b32\w*\s+v(\d+),\s+v(\d+)\s*$", lines[j])
            copied = mv is not None and int(mv.group(2)) in dst
            w = re.match(r"v_\w+\s+v\[?(\d+)(?::(\d+))?\]?", lines[j])      # result overwritten: stop tracking those registers
            if w:
                dst -= set(range(int(w.group(1)), int(w.group(2) or w.group(1)) + 1))
            if copied:
                dst.add(int(mv.group(1)))                              # ... but a copy of a tracked register is tracked too
    return n


def isa_lines(path):
    return [l.strip() for l in open(path) if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]


def _compile(src, force, extra):
    import glob
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".", "_") + ".o")
    deps = [path] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _newer(obj, deps):
        return obj, None
    device = src.endswith(".hip")
    cmd = [HIPCC, *COMMON, *flags_for(src), *extra, *(["-save-temps=obj"] if device else []), "-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if device:          # the device ISA came out of the same compilation (-save-temps): scan it, then drop the temporaries
        stem = os.path.join(OBJ, os.path.splitext(src)[0])
        asm = stem + "-hip-amdgcn-amd-amdhsa-gfx950.s"
        pairs = count_pk_f64_pairs(isa_lines(asm)) if os.path.exists(asm) else -1
        for f in glob.glob(stem + "-hip-amdgcn-amd-amdhsa-gfx950.*") + glob.glob(stem + "-host-x86_64-unknown-linux-gnu.*") + \
                glob.glob(stem + ".hip-hip-amdgcn-amd-amdhsa.hipfb"):
            os.remove(f)
        if pairs != 0:
            os.remove(obj)
            raise RuntimeError(f"{src}: " + ("no device ISA was produced to scan" if pairs < 0 else
                               f"{pairs} packed-fp32 -> v_cvt_f64_f32 pair(s) in the gfx950 ISA (a hazard on MI355X: see NO_SLP above); "
                               "build the source with -fno-slp-vectorize or form the product in float32 first"))
    return obj, r.stderr


def build(force=False, verbose=False, extra=()):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, list(extra)), SOURCES))
    objs = [o for o, _ in results]
    if verbose:
        for (_, log), s in zip(results, SOURCES):
            if log:
                print(f"--- {s}\n{log}")
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, verbose="-v" in sys.argv,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
    print(lib)
