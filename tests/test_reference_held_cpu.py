"""The only numbers the reference itself holds for this path (tests/golden/reference_held.json), on the CPU:
its committed notebook run logs the seed, the initial best point and the hyper-parameters of its first GP fit.
The Sobol design is seeded NumPy + SciPy, the likelihood is a closed form, and the fit is scipy L-BFGS-B from the
x0 recipe of pool.py:277-284 — so the oracle must land on the logged digits.  This pins (for N = 2, d = 2):
scale_to_unit / scale_from_unit, y standardisation, the RBF kernel, gp_mll, the analytic gradient, the default
priors and log-bounds, the restart recipe and optimize_scipy's screening / acceptance rule against the reference's
own JAX run.  (It is a small pin: everything else in tests/golden is the oracle's output, not the reference's.)"""
import json
import os

import numpy as np
from scipy.stats import qmc

from oracle import bobe_oracle as O

HELD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_held.json")))["notebook_banana"]


def banana(x):       # the likelihood text recorded next to the held values
    return -0.25 * (5 * (0.2 - x[0])) ** 2 - (20 * (x[1] / 4 - x[0] ** 4)) ** 2


def notebook_initial_design():
    """bo.py:505-537 with the logged seed: returns (rng after the draw, unit-cube points, physical points, values)."""
    rng = np.random.default_rng(HELD["seed"])
    bounds = np.array(HELD["param_bounds"]).T                                  # (2, ndim) like the reference
    unit = qmc.Sobol(d=2, scramble=True, seed=rng).random(max(2, HELD["constructor"]["n_sobol_init"]))
    phys = bounds[0] + unit * (bounds[1] - bounds[0])                          # utils/core.py:188-193
    vals = np.array([banana(p) for p in phys]).reshape(-1, 1)
    return rng, unit, phys, vals


def test_initial_design_reproduces_the_logged_best_point():
    _, _, phys, vals = notebook_initial_design()
    i = int(np.argmax(vals))
    got = {n: f"{v:.6f}" for n, v in zip(HELD["param_list"], phys[i])}         # the reference's log format (bo.py:412)
    assert got == HELD["logged_initial_best_point"]
    assert f"{vals[i, 0]:.6f}" == HELD["logged_initial_best_value"]


def test_oracle_fit_lands_on_the_logged_hyperparameters():
    rng, unit, _, vals = notebook_initial_design()
    gp = O.OracleGP(unit, vals)                                                # reference defaults (bo.py:584-605, gp_kwargs={})
    before = {"lengthscales": {n: f"{v:.4f}" for n, v in zip(HELD["param_list"], gp.lengthscales)},
              "kernel_variance": f"{gp.kernel_variance:.4f}"}
    assert before == HELD["logged_hyperparameters_before_refit"]
    x0 = O.restart_points(np.log(gp.get_hyperparams()), gp.hyperparam_bounds, 4, rng)      # pool.py:277-286
    res = gp.fit(x0=x0, maxiter=500)                                           # bo.py:611
    gp.update_hyperparams(res["params"])
    after = {"lengthscales": {n: f"{v:.4f}" for n, v in zip(HELD["param_list"], gp.lengthscales)},
             "kernel_variance": f"{gp.kernel_variance:.4f}"}
    assert after == HELD["logged_hyperparameters_after_refit"]
