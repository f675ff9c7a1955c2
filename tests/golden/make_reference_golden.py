"""Pins parity: writes tests/golden/ref_*.npz from the REFERENCE ITSELF (Ameek94/BOBE with jax / numpyro installed).

This script cannot run in the build container (``import BOBE.gp`` fails there with ModuleNotFoundError: jax), which is
why the committed fixtures come from the CPU oracle and the parity status is "unpinned" (DESIGN.md 2).  On any machine
that has the reference's environment (environment.yml: jax 0.5.3, numpyro 0.15.3):

    pip install -e /path/to/BOBE && python tests/golden/make_reference_golden.py

produces files with exactly the keys of make_golden.py for the same seeded inputs.  tests/test_golden_cpu.py and
tests/test_gpu_parity.py pick every tests/golden/*.npz up automatically: committing the ref_*.npz files turns both the
oracle check and the HIP check into checks against the reference's own outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import CASES  # noqa: E402  (same cases, same seeded recipes)


def make(name):
    import jax
    jax.config.update("jax_enable_x64", True)
    import jax.numpy as jnp
    from BOBE.acquisition import EI, LogEI, WIPStd, WIPV
    from BOBE.gp import GP

    n, d, kernel, prior, M, C, seed = CASES[name]
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(n, d))
    y = -np.sum((X - 0.5) ** 2, axis=1).reshape(-1, 1) + 0.05 * np.sin(7 * X[:, :1])
    ls = 0.3 + 0.1 * np.arange(d)
    gp = GP(train_x=X, train_y=y, noise=1e-6, kernel=kernel, lengthscales=jnp.array(ls), kernel_variance=1.5,
            lengthscale_prior=prior)
    theta = np.log(np.asarray(gp.get_hyperparams())) + 0.07 * np.cos(np.arange(gp.num_hyperparams))
    f, g = jax.value_and_grad(gp.neg_mll)(jnp.array(theta))              # what optim.py:306-309 evaluates
    cand = rng.uniform(0, 1, size=(C, d))
    cand[1] = X[4]
    Z = rng.uniform(0, 1, size=(M, d))
    k_train_mc = gp.kernel(gp.train_x, jnp.array(Z), gp.lengthscales, gp.kernel_variance, gp.noise, include_noise=False)
    wipv_f, wipstd_f = WIPV(), WIPStd()
    wipv = np.array([float(wipv_f.fun(jnp.array(c), gp, mc_points=jnp.array(Z), k_train_mc=k_train_mc)) for c in cand])
    wipstd = np.array([float(wipstd_f.fun(jnp.array(c), gp, mc_points=jnp.array(Z), k_train_mc=k_train_mc)) for c in cand])
    fant = np.array([np.asarray(gp.fantasy_var(jnp.array(c), jnp.array(Z), k_train_mc)) for c in cand[:6]])
    mean_std, var_std = gp.predict_batched(jnp.array(cand))              # standardised, floors applied
    best = float(np.max(np.asarray(gp.train_y)))
    ei = np.array([-float(EI().fun(jnp.array(c), gp, best, 0.0)) for c in cand])
    log_ei = np.array([-float(LogEI().fun(jnp.array(c), gp, best, 0.0)) for c in cand])
    return dict(X=X, y=y, lengthscales=ls, kernel_variance=1.5, noise=1e-6, kernel=kernel,
                prior="none" if prior is None else prior, theta=theta, neg_mll=float(f), neg_mll_grad=np.asarray(g),
                cholesky=np.asarray(gp.cholesky), alphas=np.asarray(gp.alphas), cand=cand, Z=Z,
                mean=np.asarray(mean_std), var=np.asarray(var_std), wipv=wipv, wipstd=wipstd,
                argmin_v=int(np.argmin(wipv)), argmin_s=int(np.argmin(wipstd)), fantasy=fant,
                pred_mean=np.asarray(gp.predict_mean_batched(jnp.array(cand))),
                pred_var=np.asarray(gp.predict_var_batched(jnp.array(cand))), ei=ei, log_ei=log_ei, best_y=best,
                y_mean=float(gp.y_mean), y_std=float(gp.y_std))


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, "ref_" + name + ".npz"), **make(name))
        print("wrote ref_" + name)
