"""Seeded synthetic workload of the headline benchmark (SURVEY.md 8d / BASELINE.md section 2).

Shared by bench.py and the parity tests so that the GPU path and the CPU oracle see identical inputs.
Data generation only — not part of the measured path."""
from __future__ import annotations

import math

import numpy as np

CONFIGS = {
    # name: (N, d, C, M)
    "tiny": (256, 4, 1024, 128),
    "small": (1024, 6, 8192, 512),
    "headline": (4096, 8, 65536, 512),
    # not in BASELINE.json: a 4x larger training set with twice the candidates, to show where the engine goes with
    # size (the synthetic prior draw needs an O(N^3) CPU Cholesky: ~20 s at this N)
    "large": (8192, 8, 131072, 512),
}


def synthetic_problem(N: int, d: int, C: int, M: int = 512, ls_star: float = 0.6, noise: float = 1e-6,
                      cand_offset: int = 0):
    """X ~ U[0,1]^(N x d) (default_rng(1234)); y = one draw from the RBF GP prior (ls*=0.6, kvar=1,
    noise=1e-6), standardised; candidates = scrambled Sobol(seed 5678) rows [cand_offset, cand_offset+C);
    Z = first M rows of scrambled Sobol(seed 9012)."""
    from scipy.linalg import cholesky
    from scipy.stats import qmc
    rng = np.random.default_rng(1234)
    X = rng.uniform(0.0, 1.0, (N, d))
    Xs = X / ls_star
    sq = np.zeros((N, N))
    for j in range(d):
        df = Xs[:, j][:, None] - Xs[:, j][None, :]
        sq += df * df
    K = np.exp(-0.5 * sq) + noise * np.eye(N)
    L = cholesky(K, lower=True, check_finite=False)
    y = L @ rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    cand = sobol_candidates(d, C, cand_offset)
    Z = qmc.Sobol(d, scramble=True, seed=9012).random(M)
    return X, y, cand, Z


def sobol_candidates(d: int, C: int, offset: int = 0) -> np.ndarray:
    """rows [offset, offset + C) of the candidate set: scrambled Sobol, seed 5678 (a rank generates its own shard)"""
    from scipy.stats import qmc
    sob = qmc.Sobol(d, scramble=True, seed=5678)
    if offset:
        sob.fast_forward(offset)
    return sob.random(C) if C > 0 else np.empty((0, d))


def theta_schedule(d: int, ls_star: float = 0.6, n: int = 20) -> np.ndarray:
    """theta_k = log ls* + 0.05 k (-1)^k for k = 0..n-1 (all dims), log kvar = 0: a fixed schedule so
    that CPU and GPU do identical work regardless of the optimiser path."""
    th = np.empty((n, d + 1))
    for k in range(n):
        th[k, :d] = math.log(ls_star) + 0.05 * k * (-1) ** k
        th[k, d] = 0.0
    return th
