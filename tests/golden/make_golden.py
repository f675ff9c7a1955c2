"""Generates tests/golden/*.npz — small seeded input/output vectors for the GP hot path.

The reference (Ameek94/BOBE) cannot be imported in the build container (jax/numpyro absent) and its
tests contain no numeric vectors, so these fixtures are produced by the CPU oracle
(oracle/bobe_oracle.py, "parity unpinned") after tests/test_oracle.py has pinned it against independent
implementations.  They freeze today's oracle outputs so that later changes to the oracle or the HIP
path are caught by both the CPU suite (oracle vs fixture) and the GPU suite (HIP vs fixture).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bobe_oracle as O  # noqa: E402

CASES = {
    # name: (n, d, kernel, prior, M, C, seed)
    "rbf_n50_d2": (50, 2, "rbf", None, 16, 40, 42),
    "matern_n130_d3": (130, 3, "matern", "DSLP", 24, 50, 7),
    "rbf_n257_d5_saas": (257, 5, "rbf", "SAAS", 32, 64, 3),
}


def make(name):
    n, d, kernel, prior, M, C, seed = CASES[name]
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(n, d))                      # reference tests/test_gp.py:21-27 recipe
    y = -np.sum((X - 0.5) ** 2, axis=1).reshape(-1, 1) + 0.05 * np.sin(7 * X[:, :1])
    ls = 0.3 + 0.1 * np.arange(d)
    gp = O.OracleGP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.5, lengthscale_prior=prior)
    theta = np.log(gp.get_hyperparams()) + 0.07 * np.cos(np.arange(gp.num_hyperparams))
    f, g = gp.neg_mll_value_and_grad(theta)
    cand = rng.uniform(0, 1, size=(C, d))
    cand[1] = X[4]
    Z = rng.uniform(0, 1, size=(M, d))
    sw = O.wip_sweep(gp, cand, Z)
    fant = np.array([gp.fantasy_var(c, Z, gp._k12(Z)) for c in cand[:6]])
    best = float(np.max(gp.train_y))
    return dict(X=X, y=y, lengthscales=ls, kernel_variance=1.5, noise=1e-6, kernel=kernel,
                prior="none" if prior is None else prior, theta=theta, neg_mll=f, neg_mll_grad=g,
                cholesky=gp.cholesky, alphas=gp.alphas, cand=cand, Z=Z, mean=sw["mean"], var=sw["var"],
                wipv=sw["wipv"], wipstd=sw["wipstd"], argmin_v=sw["argmin_v"], argmin_s=sw["argmin_s"],
                fantasy=fant, pred_mean=gp.predict_mean_batched(cand), pred_var=gp.predict_var_batched(cand),
                ei=O.ei_score(sw["mean"], sw["var"], best), log_ei=O.log_ei_score(sw["mean"], sw["var"], best),
                best_y=best, y_mean=gp.y_mean, y_std=gp.y_std)


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **make(name))
        print("wrote", name)
