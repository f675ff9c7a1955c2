"""Cost of one value+gradient evaluation at BO-loop sizes (N = 20 ... 600): alone, and 4 / 8 at a time (slots)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402

for N, d in ((20, 2), (60, 2), (150, 2), (300, 10), (600, 10)):
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    gp = GP(X, -np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.6))
    ls = np.full((8, d), 0.55) + 0.01 * np.arange(8)[:, None]
    out = f"N={N:4d} d={d:2d}:"
    for B in (1, 4, 8):
        f = (lambda: gp.mll_data(ls[0], 1.0)) if B == 1 else (lambda: gp.mll_data_batch(ls[:B], np.ones(B)))
        f()
        t0 = time.perf_counter()
        for _ in range(200):
            f()
        dt = (time.perf_counter() - t0) / 200
        out += f"  B={B}: {dt*1e6:7.1f} us/call = {dt*1e6/B:6.1f} us/eval"
    print(out, flush=True)
