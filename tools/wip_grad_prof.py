"""rocprofv3 --kernel-trace target: 50 bobe_gp_wip_grad calls at N=600 d=10 M=256 C=1 (tools/kstats-like summary by tools/wip_grad_prof.py parse)."""
import collections
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from bobe_amd.gp import GP
    N, d, M = 600, 10, 256
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    gp = GP(X, -np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.6))
    Z = rng.uniform(size=(M, d))
    c = rng.uniform(size=(1, d))
    for i in range(50):
        c[0, 0] = 0.3 + 1e-3 * i
        gp.wip_grad(c, Z)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", "")
        tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[n] += 1
    for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"{n[:60]:60s} {cnt[n]:5d} launches {v:10.1f} us  avg {v / cnt[n]:9.2f} us")
