"""MFMA utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).

util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
is summed over the 8 XCDs; MI355X_MICROARCH.md 'DVFS give-back').  SQ_VALU_MFMA_BUSY_CYCLES counts cycles.
    python tools/pmc_mfma.py <dir> > profiles/r01_mfma_util.json
"""
import csv
import glob
import json
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {}
for k, v in acc.items():
    if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        out[k] = {"dispatches": cnt[k], "mfma_util": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                  "mean_kernel_cycles": cyc / max(cnt[k], 1)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"])), indent=1))
