"""GP.fit wall time per evaluation for the three restart drivers (sequential / slots+threads / lock-step batch) vs N."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import GP  # noqa: E402
from bobe_amd.optim import optimize_scipy  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or (64, 256, 512, 1024, 2048):
    rng = np.random.default_rng(0)
    d = 4
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1))
    gp = GP(X, y, noise=1e-6, lengthscale_bounds=[0.05, 5], kernel_variance_bounds=[1e-2, 1e2])
    x0 = np.vstack([np.log(gp.get_hyperparams()), rng.uniform(gp.hyperparam_bounds[0], gp.hyperparam_bounds[1], size=(3, d + 1))])
    calls = [0]
    orig = gp.mll_data
    orig_b = gp.mll_data_batch

    def counted(*a, **k):
        calls[0] += 1
        return orig(*a, **k)

    def counted_b(ls, kv, *a, **k):
        calls[0] += len(kv)
        return orig_b(ls, kv, *a, **k)
    gp.mll_data, gp.mll_data_batch = counted, counted_b
    out = []
    for mode in ("sequential", "slots", "batch"):
        kw = {}
        if mode == "slots":
            kw = {"slot_value_and_grad": lambda x, slot: gp.neg_mll_value_and_grad(x, slot=slot), "n_slots": 4}
        elif mode == "batch":
            kw = {"batch_value_and_grad": gp.neg_mll_value_and_grad_batch}
        for rep in range(2):
            calls[0] = 0
            t0 = time.perf_counter()
            r = optimize_scipy(gp.neg_mll_value_and_grad, num_params=d + 1, bounds=gp.hyperparam_bounds, x0=x0, maxiter=60,
                               n_restarts=4, optimizer_options={}, **kw)
            dt = time.perf_counter() - t0
        out.append(f"{mode}: {dt * 1e3:7.1f} ms / {calls[0]} evals = {dt * 1e6 / calls[0]:6.0f} us each (f={r[1]:.6f})")
    print(f"N={N}: " + " | ".join(out), flush=True)
