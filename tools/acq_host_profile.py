"""Host-side profile of one kriging-believer batch (WIPStd, batch of 5, M = 256 integration points) at a BO-loop size."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.acquisition import WIPStd
from bobe_amd.gp import GP
N, d = int(os.environ.get("PROBE_N", "400")), 10
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1.0 - X[:, :-1]) ** 2, axis=1) / 20.0
gp = GP(X, y, noise=1e-8, lengthscales=np.full(d, 0.5), kernel_variance=3.0)
mc = {"x": np.random.default_rng(1).uniform(size=(2048, d))}
acq = WIPStd()
kw = dict(n_batch=5, acq_kwargs={"mc_samples": mc, "mc_points_size": 256}, n_restarts=1, maxiter=100, early_stop_patience=10)
acq.get_next_batch(gp, rng=np.random.default_rng(2), **kw)
t0 = time.perf_counter()
for i in range(5):
    acq.get_next_batch(gp, rng=np.random.default_rng(3 + i), **kw)
print(f"N={N}: {(time.perf_counter() - t0) * 200:.2f} ms per believer batch of 5")
pr = cProfile.Profile(); pr.enable()
acq.get_next_batch(gp, rng=np.random.default_rng(9), **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
