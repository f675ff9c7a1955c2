"""Concurrency analysis of a rocprofv3 kernel trace: per-stream kernel time, union busy time, overlap histogram.
usage: python tools/trace_overlap.py <kernel_trace.csv> [min_streams]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), r["Kernel_Name"].split("(")[0][-28:]) for r in rows]
ev.sort()
# find windows where >= 3 distinct streams are active within 2 ms: take the region of the trace where streams != main
streams = collections.Counter(e[2] for e in ev)
print("streams:", dict(streams))
batch_streams = [s for s, c in streams.items() if s not in (0, 1)]
be = [e for e in ev if e[2] in batch_streams]
if not be:
    sys.exit("no batch streams")
# split into bursts separated by gaps > 200 us
bursts, cur = [], [be[0]]
for e in be[1:]:
    if e[0] - max(x[1] for x in cur[-50:]) > 200_000:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
print("bursts:", len(bursts))
b = bursts[len(bursts) // 2]
t0, t1 = min(e[0] for e in b), max(e[1] for e in b)
print(f"burst: {len(b)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
per = collections.defaultdict(float)
cls = collections.defaultdict(float)
for s, e, st, n in b:
    per[st] += (e - s) / 1e6
    cls[n] += (e - s) / 1e6
print("kernel time per stream (ms):", {k: round(v, 3) for k, v in per.items()})
for n, v in sorted(cls.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {n:30s} {v:8.3f} ms")
# overlap histogram
pts = []
for s, e, st, n in b:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
hist = collections.defaultdict(float)
lvl, last = 0, pts[0][0]
for t, dlt in pts:
    hist[lvl] += (t - last) / 1e6
    last = t
    lvl += dlt
print("time with k kernels in flight (ms):", {k: round(v, 3) for k, v in sorted(hist.items())})
