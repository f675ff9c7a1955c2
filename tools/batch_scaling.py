"""Wall time of bobe_gp_mll_batch vs the number of concurrent evaluations (GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 8
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
y = np.sin(X.sum(1))
gp = GP(X, y, noise=1e-4, lengthscales=np.full(d, 0.6))
for B in (1, 2, 3, 4, 6, 8):
    ls = np.full((B, d), 0.55) + 0.01 * np.arange(B)[:, None]
    kv = np.ones(B)
    gp.mll_data_batch(ls, kv)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gp.mll_data_batch(ls, kv)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={N} B={B}: {dt * 1e3:.3f} ms per batch, {dt * 1e3 / B:.3f} ms per evaluation", flush=True)
