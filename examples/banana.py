"""BASELINE.json config 1 — 2-D curved ("banana") likelihood, reference examples/Banana.py:14-18, 27-31, 54-68,
end to end on the GPU GP: WIPStd with HMC integration points and the logZ convergence test on the surrogate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402


def loglike(X):
    return -0.25 * (5 * (0.2 - X[0])) ** 2 - (20 * (X[1] / 4 - X[0] ** 4)) ** 2


def true_logz(bounds, n=2001):
    """log of the prior-averaged likelihood (uniform prior on the box) by the trapezoid rule."""
    x = np.linspace(bounds[0, 0], bounds[1, 0], n)
    y = np.linspace(bounds[0, 1], bounds[1, 1], n)
    L = np.exp(-0.25 * (5 * (0.2 - x[:, None])) ** 2 - (20 * (y[None, :] / 4 - x[:, None] ** 4)) ** 2)
    integral = np.trapezoid(np.trapezoid(L, y, axis=1), x)
    return float(np.log(integral / ((x[-1] - x[0]) * (y[-1] - y[0]))))


if __name__ == "__main__":
    bounds = np.array([[-1, 1], [-1, 2]]).T
    t0 = time.time()
    bobe = BOBE(loglike, ["x1", "x2"], bounds, n_sobol_init=8, seed=42, save=False)
    # the run settings of the reference's examples/Banana.py:53-68 (HMC integration points, logZ convergence on
    # the surrogate, final nested sampling); MAX_EVALS shortens the run
    max_evals = int(os.environ.get("MAX_EVALS", 250))
    res = bobe.run(acq="wipstd", min_evals=25, max_evals=max_evals, max_gp_size=max_evals, logz_threshold=5e-2,
                   do_final_ns=True, fit_n_points=1, batch_size=1, ns_n_points=1, num_hmc_warmup=512,
                   num_hmc_samples=2048, mc_points_size=512, num_chains=4, convergence_n_iters=2)
    print("termination:", res["termination_reason"], "| logZ:", {k: round(float(v), 3) for k, v in res["logz"].items()
                                                                   if k in ("mean", "upper", "lower")},
          "| direct quadrature: %.3f" % true_logz(bounds))
    print("banana: %d evals in %.1fs, best logL %.4f at %s" % (res["n_evals"], time.time() - t0, res["best_val"], res["best_x"]))
    print("timing:", {k: round(v, 2) for k, v in res["timing"].items()})
