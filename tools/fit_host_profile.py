"""Where a small-N fit spends its HOST time (the BO loop's regime: N = 100 ... 1200): cProfile of GP.fit with the restarts
one after the other (no threads, so the profile is readable) and wall times of the three drivers."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import gp_fit  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402

N, d = int(os.environ.get("PROBE_N", "600")), 10
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1.0 - X[:, :-1]) ** 2, axis=1) / 20.0
gp = GP(X, y, noise=1e-8)
if os.environ.get("SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SWITCH_INTERVAL"]))
for mode, conc in (("sequential", False), ("slots", True), ("lockstep", True)):
    gp.concurrent_restarts, gp.restart_mode = conc, ("lockstep" if mode == "lockstep" else "slots")
    gp.update_hyperparams(np.zeros(d + 1))
    gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)
    gp.update_hyperparams(np.zeros(d + 1))
    calls = [0]
    orig, origb = gp.mll_data, gp.mll_data_batch

    def c1(*a, **k):
        calls[0] += 1
        return orig(*a, **k)

    def cb(ls, kv, *a, **k):
        calls[0] += len(kv)
        return origb(ls, kv, *a, **k)
    gp.mll_data, gp.mll_data_batch = c1, cb
    t0 = time.perf_counter()
    res = gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)
    dt = time.perf_counter() - t0
    gp.mll_data, gp.mll_data_batch = orig, origb
    print(f"N={N} {mode:10s}: {dt * 1e3:7.1f} ms, {calls[0]} evaluations, {dt * 1e6 / calls[0]:.0f} us per evaluation; mll {res['mll']!r}", flush=True)
if os.environ.get("NO_PROFILE"):
    sys.exit(0)
gp.concurrent_restarts = False
gp.update_hyperparams(np.zeros(d + 1))
pr = cProfile.Profile()
pr.enable()
gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
