"""Per-kernel means of whatever counters a rocprofv3 --pmc pass collected (counter_collection.csv in DIR)."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", "")
    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
keep = sys.argv[2:] or None
for (k, c), v in sorted(acc.items()):
    if keep and not any(s in k for s in keep):
        continue
    print(f"{k[:44]:44s} {c:22s} n={len(v):5d} mean={sum(v) / len(v):14.3f}")
