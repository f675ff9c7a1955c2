"""Kernel sequence of ONE value+gradient evaluation at a BO-loop size, from a rocprofv3 kernel trace.
  run  : BOBE_GRAPH_MAX_N=0 rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/small_n_trace.py run N d
  parse: python tools/small_n_trace.py parse <kernel_trace.csv>"""
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from bobe_amd.gp import GP
    N, d = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    gp = GP(X, -np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.6))
    for i in range(5):
        gp.mll_data(np.full(d, 0.55 + 0.01 * i), 1.0)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", ""))
                for r in rows)
    last = max(i for i, e in enumerate(ev) if "k_scale_coords" in e[2])
    seg = ev[last:]
    t0 = seg[0][0]
    prev = None
    for s, e, n in seg:
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{(s - t0) / 1e3:8.1f} us  +{gap:5.1f} gap  {(e - s) / 1e3:6.1f} us  {n[:50]}")
        prev = e
    print(f"span {(seg[-1][1] - t0) / 1e3:.1f} us over {len(seg)} kernels")
