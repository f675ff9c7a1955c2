import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bobe_amd.gp import GP
from bobe_amd.bo import gp_fit
from bobe_amd.synthetic import synthetic_problem
"""GP.fit (4 restarts, L-BFGS-B, maxiter 200) with the restarts on evaluation slots vs in lock step, warm."""
for N in (600, 1024, 2048, 4096):
    X, y, _, _ = synthetic_problem(N, 8, 8, 8, noise=1e-6)
    for mode in ("slots", "lockstep", "slots", "lockstep"):
        gp = GP(X, y, noise=1e-6, lengthscales=np.full(8, 0.6))
        gp.restart_mode = mode
        gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)   # warm: workspaces exist
        gp.update_hyperparams(np.log(np.append(np.full(8, 0.6), 1.0)))
        calls = [0]; orig = gp.mll_data; origb = gp.mll_data_batch
        def c1(*a, **k): calls[0] += 1; return orig(*a, **k)
        def cb(ls, kv, *a, **k): calls[0] += len(kv); return origb(ls, kv, *a, **k)
        gp.mll_data = c1; gp.mll_data_batch = cb
        t0 = time.perf_counter()
        r = gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)
        dt = time.perf_counter() - t0
        print(f"N={N} {mode:9s}: {dt*1e3:8.1f} ms, {calls[0]} evaluations, {dt*1e3/calls[0]:.3f} ms/eval, mll {r['mll']:.6f}", flush=True)
