"""An 8-restart fit (the BO loop's policy below 200 points, bo.py:639-653) against the number of evaluations in flight."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import gp_fit
from bobe_amd.gp import GP
d = 10
for N in (64, 150, 400, 900):
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1.0 - X[:, :-1]) ** 2, axis=1) / 20.0
    gp = GP(X, y, noise=1e-8)
    for rep in range(2):
        gp.update_hyperparams(np.zeros(d + 1))
        t0 = time.perf_counter()
        r = gp_fit(gp, maxiters=1000, n_restarts=8, rng=np.random.default_rng(7), distributed=False)
        dt = time.perf_counter() - t0
    print(f"BOBE_MLL_SLOTS={os.environ.get('BOBE_MLL_SLOTS', '4')} N={N}: 8-restart fit {dt * 1e3:.1f} ms, mll {r['mll']:.6f}", flush=True)
