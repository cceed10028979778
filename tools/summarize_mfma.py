#!/usr/bin/env python3
"""Fold a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
--kernel-trace` pass of the serial bench command (RFI_NO_OVERLAP=1) into matrix-pipe utilisation and
effective shader clock of the two persistent MFMA kernel families.

    python tools/summarize_mfma.py gpurun_out/pmc_mfma_r1 profiles/r1_mfma_util.json

Both kernels are persistent (every wave lives for the whole launch), so per dispatch
kernel_cycles = 4 x SQ_WAVE_CYCLES / n_waves (SQ_WAVE_CYCLES counts quad-cycles summed over waves),
eff_clock = kernel_cycles / duration, and mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
kernel_cycles).  (v_mfma_f32_32x32x2_f32 keeps a SIMD's pipe busy 64 cycles per instruction, v_mfma_f32_32x32x16_bf16 32.)
(GRBM_GUI_ACTIVE is not used: it runs ~8 us longer than the dispatch and not at the shader clock.)"""
import collections
import csv
import glob
import json
import sys

from summarize_pmc import family

d, out = sys.argv[1:3]
f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Dispatch_Id"]
    disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[k]["name"] = r["Kernel_Name"]
    disp[k]["waves"] = float(r["Grid_Size"]) / 64.0
    disp[k]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
import re

acc = collections.defaultdict(lambda: collections.defaultdict(float))
for v in disp.values():
    fam = family(v["name"])
    if fam not in ("conv_igemm_mfma", "wgrad_igemm_mfma"):
        continue
    cyc = 4.0 * v["SQ_WAVE_CYCLES"] / v["waves"]
    short = re.sub(r"rfi::\(anonymous namespace\)::|void |\(.*$", "", v["name"])
    for key in (fam, "kernel:" + short):
        a = acc[key]
        a["n"] += 1
        a["ns"] += v["ns"]
        a["cycles"] += cyc
        a["mfma"] += v["SQ_VALU_MFMA_BUSY_CYCLES"]
res, kern = {}, {}
for key, a in acc.items():
    row = {"launches_seen": int(a["n"]), "avg_launch_us": round(a["ns"] / a["n"] / 1e3, 2),
           "eff_clock_GHz": round(a["cycles"] / a["ns"], 3),
           "mfma_util": round(a["mfma"] / (1024.0 * a["cycles"]), 4)}
    (kern if key.startswith("kernel:") else res)[key.replace("kernel:", "")] = row
json.dump({"families": res, "kernels": kern, "note": " ".join(__doc__.split("\n\n")[-1].split())}, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
