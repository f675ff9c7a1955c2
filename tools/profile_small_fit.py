"""Where a small-N fit spends its time (GPU box): cProfile of GP.fit with 8 restarts at N = 60."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import GP  # noqa: E402
from bobe_amd.bo import gp_fit  # noqa: E402

rng = np.random.default_rng(0)
X = rng.uniform(size=(60, 2))
y = -np.sum((X - 0.5) ** 2, axis=1)
gp = GP(X, y, noise=1e-8, lengthscale_bounds=[0.01, 10], kernel_variance_bounds=[1e-4, 1e8])
calls = [0]
orig = gp.mll_data


def counted(*a, **k):
    calls[0] += 1
    return orig(*a, **k)


gp.mll_data = counted
for conc in (True, True, False):     # the first concurrent fit pays the one-time slot set-up (~0.8 s)
    gp.concurrent_restarts = conc
    calls[0] = 0
    t0 = time.perf_counter()
    gp_fit(gp, n_restarts=8, maxiters=1000, rng=np.random.default_rng(1))
    dt = time.perf_counter() - t0
    print(f"concurrent={conc}: {dt * 1e3:.1f} ms, {calls[0]} evaluations, {dt * 1e6 / max(calls[0], 1):.0f} us each", flush=True)
gp.concurrent_restarts = False
pr = cProfile.Profile()
pr.enable()
gp_fit(gp, n_restarts=8, maxiters=1000, rng=np.random.default_rng(1))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
