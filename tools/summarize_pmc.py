#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel-family HBM traffic per launch.

    python tools/summarize_pmc.py gpurun_out/pmc_fetch_r1 gpurun_out/pmc_write_r1 profiles/r1_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are
in KiB; on gfx950 FETCH_SIZE tallies each 128-byte request of a wide coalesced read as 64 bytes, so
the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import re
import sys

FAMILY = [(r"conv_igemm_kernel|pconv_kernel|conv_ws_kernel|gemm_ws_kernel|conv_stem_kernel", "conv_igemm_mfma"),
          (r"wgrad_igemm_kernel|pwgrad_kernel|wgrad_split_kernel|wgrad_ws_kernel", "wgrad_igemm_mfma"),
          (r"bn_|finish_channel_sum", "batchnorm"), (r"reduce_slabs", "slab_reduce"),
          (r"adam|sumsq", "optimizer")]


def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam
    return "elementwise" if "rfi::" in name else None


def short(name):
    """`void rfi::(anonymous namespace)::conv_ws_kernel<1, 8, 32, 2, 0, false, 3>(rfi::...)` -> `conv_ws_kernel<1, 8, 32, 2, 0, false, 3>`"""
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"rfi::\(anonymous namespace\)::|rfi::", "", name)
    depth, out = 0, []
    for ch in name:                      # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)


def fold(d):
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam:
            tot[fam] += float(r["Counter_Value"])
            n[fam] += 1
            k = "kernel:" + short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    ft, fn = fold(fetch_dir)
    wt, wn = fold(write_dir)
    res = {"note": "bytes per launch; fetch = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction), "
                   "write = WRITE_SIZE x 1024; rocprofv3 --kernel-trace --pmc, separate passes",
           "families": {}}
    res["kernels"] = {}
    for fam in ft:
        fetch = 2.0 * ft[fam] * 1024 / fn[fam]
        write = wt.get(fam, 0.0) * 1024 / max(wn.get(fam, 1), 1)
        rec = {"launches_seen": fn[fam], "fetch_bytes_per_launch": round(fetch),
               "write_bytes_per_launch": round(write), "hbm_bytes_per_launch": round(fetch + write)}
        if fam.startswith("kernel:"):
            res["kernels"][fam[7:]] = rec
        else:
            res["families"][fam] = rec
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
