"""Cost of one bobe_gp_wip_grad call as the L-BFGS refinement of an acquisition point issues it (C = 1, one Z)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402

for N, d, M in ((100, 2, 128), (600, 10, 256), (2000, 10, 512)):
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    gp = GP(X, -np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.6))
    Z = rng.uniform(size=(M, d))
    for C in (1, 64):
        c = rng.uniform(size=(C, d))
        gp.wip_grad(c, Z)
        t0 = time.perf_counter()
        for i in range(200):
            c[0, 0] = 0.3 + 1e-4 * i
            gp.wip_grad(c, Z)
        dt = (time.perf_counter() - t0) / 200
        print(f"N={N} d={d} M={M} C={C}: {dt*1e6:8.1f} us per call", flush=True)
