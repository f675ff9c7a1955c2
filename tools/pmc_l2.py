"""L2 hit rate per kernel from one rocprofv3 PMC pass (TCC_HIT_sum, TCC_MISS_sum).
    python tools/pmc_l2.py <dir>"""
import csv
import glob
import json
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in acc.items():
    h, m = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    if h + m > 0:
        out[k] = {"l2_hit_rate": h / (h + m), "requests": h + m}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["requests"])[:12]), indent=1))
