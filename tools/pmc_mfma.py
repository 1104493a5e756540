#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (counter_collection.csv):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python bench.py ...
util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs): the fraction of all matrix-pipe cycles of the chip that
were busy while the kernel ran.  Kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs,
MI355X_MICROARCH.md "DVFS give-back").  usage: pmc_mfma.py <dir> [out.json]"""
import csv, glob, json, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
N_CU, SIMD = 256, 4
out = {}
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    util = mf / (gui / 8.0 * N_CU * SIMD)
    out[k] = {"launches": cnt[k], "mfma_busy_cycles": mf, "gui_active_cycles": gui, "mfma_util": util}
    print(f"{k:70s} n={cnt[k]:5d} MFMA util {100 * util:5.1f} %")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
