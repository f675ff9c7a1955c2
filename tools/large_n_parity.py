"""Parity at sizes beyond the test suite's: L, the MLL value and its gradient at N = 8192 / 16384 against LAPACK on the host
(float64 Cholesky + the analytic gradient through the explicit inverse)."""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402
from oracle import bobe_oracle as O  # noqa: E402  (checker only)

d = 8
for N in [int(a) for a in sys.argv[1:]] or (8192, 16384):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
    ls = np.full(d, 0.6)
    t0 = time.time()
    gp = GP(X, y, noise=1e-4, lengthscales=ls, kernel_variance=1.3)
    Lg = gp.cholesky
    tg = time.time() - t0
    t0 = time.time()
    K = O.rbf_kernel(X, X, ls, 1.3, 1e-4, True)
    Lr = np.linalg.cholesky(K)
    tc = time.time() - t0
    errL = np.max(np.abs(Lg - Lr)) / np.max(np.abs(Lr))
    ys = gp.train_y.reshape(-1)
    w = sla.solve_triangular(Lr, ys, lower=True)
    mll_ref = -0.5 * w @ w - np.sum(np.log(np.diag(Lr))) - 0.5 * N * np.log(2 * np.pi)
    m, g = gp.mll_data(ls, 1.3)
    print(f"N={N}: max|L-L_lapack|/max|L| = {errL:.2e}; MLL gpu {m:.9f} lapack {mll_ref:.9f} rel {abs(m - mll_ref) / abs(mll_ref):.2e}; "
          f"GPU construct+factor+copy {tg:.2f}s, host kernel+dpotrf {tc:.2f}s", flush=True)
    del gp, K, Lr, Lg
