"""Per-kernel totals of ONE lock-step value+gradient batch (bobe_gp_mll_batch) from a rocprofv3 kernel trace.
  run   : rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/eval_kstats.py run N B
  sweep : rocprofv3 ... -- python tools/eval_kstats.py sweep N d C M      (one acquisition sweep; read with `timeline ... sweep`)
  parse : python tools/eval_kstats.py parse <kernel_trace.csv>
  timeline : python tools/eval_kstats.py timeline <kernel_trace.csv>   (every launch of the last batch: start, duration, gap)"""
import collections
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    from bobe_amd.gp import GP
    N, B = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, 8))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
    ls = np.full((B, 8), 0.55) + 0.01 * np.arange(B)[:, None]
    for _ in range(3):
        gp.mll_data_batch(ls, np.ones(B)) if B > 1 else gp.mll_data(ls[0], 1.0)
elif sys.argv[1] == "sweep":
    from bobe_amd.gp import GP
    N, d, Cn, M = (int(a) for a in sys.argv[2:6])
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(d, 0.6))
    cand, Z = rng.uniform(size=(Cn, d)), rng.uniform(size=(M, d))
    for i in range(3):
        gp.wip_sweep(cand, Z + 1e-6 * i)           # (a new Z every time: the Z-side products are part of a sweep)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", ""),
                 r.get("Stream_Id", "?")) for r in rows)
    if len(sys.argv) > 3 and sys.argv[3] == "sweep":       # the last sweep: from its Z-side coordinate scaling on
        scales = [i for i, e in enumerate(ev) if "k_scale_coords" in e[2]]
        last = scales[-2]
    else:
        last = max(i for i, e in enumerate(ev) if "k_scale_coords" in e[2] or "k_kernel_matrix" in e[2] and "k_scale_coords" not in ev[i - 1][2])
    seg = ev[last:]
    if sys.argv[1] == "timeline":
        t0, prev_end = seg[0][0], seg[0][0]
        print("   start_us   dur_us   gap_us  stream  kernel      (gap: since the end of every earlier launch)")
        for s_, e_, n, st in seg:
            print(f"{(s_ - t0) / 1e3:11.1f} {(e_ - s_) / 1e3:8.2f} {(s_ - prev_end) / 1e3:8.2f}  {st:>6}  {n}")
            prev_end = max(prev_end, e_)
        busy = sum(e_ - s_ for s_, e_, _, _ in seg) / 1e3
        print(f"span {(seg[-1][1] - t0) / 1e3:.1f} us, kernel time {busy:.1f} us, gaps {(seg[-1][1] - t0) / 1e3 - busy:.1f} us")
        sys.exit(0)
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for s, e, n, _ in seg:
        tot[n] += (e - s) / 1e3
        cnt[n] += 1
    span = (seg[-1][1] - seg[0][0]) / 1e3
    for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"{n:44s} {cnt[n]:4d} launches {v:9.1f} us  avg {v / cnt[n]:8.2f} us  {100 * v / span:5.1f} %")
    print(f"span {span:.1f} us")
