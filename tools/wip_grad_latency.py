"""Wall time per bobe_gp_wip_grad call of ONE candidate (what the L-BFGS refinement of an acquisition point issues,
acquisition.py:403-412) at BO-loop sizes, host side included.   python tools/wip_grad_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import GP  # noqa: E402

for n, d, m, noise in ((100, 10, 256, 1e-8), (300, 10, 256, 1e-8), (500, 10, 256, 1e-8), (500, 10, 256, 1e-6), (500, 2, 64, 1e-8)):
    rng = np.random.default_rng(n)
    X = rng.uniform(size=(n, d))
    y = -np.sum((X - 0.5) ** 2, axis=1) * 20
    gp = GP(X, y, noise=noise, lengthscales=np.full(d, 0.8), kernel_variance=5.0)
    Z = rng.uniform(size=(m, d))
    x = rng.uniform(size=(1, d))
    for _ in range(20):
        gp.wip_grad(x, Z)
    reps = 500
    t0 = time.perf_counter()
    for i in range(reps):
        x[0, 0] = 0.3 + 1e-4 * i
        gp.wip_grad(x, Z)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n:4d} d={d:2d} M={m:3d} noise={noise:g} refining={gp.refining}: {dt * 1e6:7.1f} us per call", flush=True)
