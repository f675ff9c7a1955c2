"""Race hunt: the same factorisation / lock-step batch / sampler launches (whole HMC chains, nested sampling's random walks,
the few-candidate score gradient) repeated under a busy GPU (a torch stream keeps the CUs occupied) must return the same
bits every time.  usage (GPU box): python tools/determinism_stress.py [reps]"""
import hashlib
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
stop = False


def noise():
    s = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda", dtype=torch.float64)
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(8):
                a = (a @ a).clamp_(-1, 1)
            s.synchronize()


th = threading.Thread(target=noise)
th.start()
bad = 0
try:
    for N, d, kern in ((300, 3, "rbf"), (1500, 8, "matern"), (2500, 5, "rbf"), (4096, 8, "rbf")):
        rng = np.random.default_rng(N)
        X = rng.uniform(size=(N, d))
        y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
        gp = GP(X, y, noise=1e-5, kernel=kern, lengthscales=np.full(d, 0.5))
        lsb = np.full((4, d), 0.4) + 0.03 * np.arange(4)[:, None]
        P = 48
        x0 = rng.uniform(0.2, 0.8, size=(P, d))
        Z = rng.uniform(size=(128, d))
        ref = None
        for r in range(reps):
            gp.recompute_cholesky()
            m, g = gp.mll_data(np.full(d, 0.45), 1.3)
            mb, gb = gp.mll_data_batch(lsb, np.ones(4))
            # the samplers' kernels: 12 adaptive HMC iterations of 48 chains, 20 random-walk steps, one score gradient
            pm, _, dm, _ = gp.predict_grad(x0, mean_only=True)
            mean = pm * gp.y_std + gp.y_mean
            state = np.ascontiguousarray(np.concatenate(
                [np.log(x0) - np.log1p(-x0), dm * gp.y_std * (x0 * (1 - x0)) + (1 - 2 * x0), x0,
                 (mean + np.sum(np.log(x0) + np.log1p(-x0), axis=1))[:, None], mean[:, None]], axis=1))
            adapt = np.tile(np.array([0.05, 0.0, 0.0, 0.0, 0.0]), (P, 1))
            gp.hmc_run(state, adapt, np.ones(d), 11, 0, 12, True, 1.0)
            xw, lw, na, ni = gp.rwalk(x0, mean, 0.03 * np.eye(d), float(np.quantile(mean, 0.3)), 20, seed=5)
            wg = gp.wip_grad(x0[:1], Z)
            h = hashlib.sha256(np.ascontiguousarray(gp.cholesky).tobytes() + np.asarray(m).tobytes() + np.asarray(g).tobytes()
                               + np.asarray(mb).tobytes() + np.asarray(gb).tobytes() + state.tobytes() + adapt.tobytes()
                               + xw.tobytes() + lw.tobytes() + na.tobytes() + b"".join(np.asarray(a).tobytes() for a in wg)
                               ).hexdigest()
            if ref is None:
                ref = h
            elif h != ref:
                bad += 1
                print(f"N={N}: repetition {r} differs", flush=True)
        print(f"N={N} {kern}: {reps} repetitions, digest {ref[:16]}", flush=True)
finally:
    stop = True
    th.join()
print("DIFFERENCES:" if bad else "all repetitions identical", bad)
sys.exit(1 if bad else 0)
