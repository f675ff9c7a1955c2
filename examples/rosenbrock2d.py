"""The reference's examples/Rosenbrock.py (2-D, bounds [-1,4] x [-1,7]) with its constructor and run settings
(Rosenbrock.py:24-66) on the GPU GP: WIPStd with HMC integration points, logZ convergence on the surrogate, a final
nested-sampling pass.  The true evidence comes from a direct quadrature of the likelihood over the box."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402


def loglike(X):
    return -((1 - X[0]) ** 2 + 100 * (X[1] - X[0] ** 2) ** 2)          # Rosenbrock.py:14-16


def true_logz(bounds, n=2001):
    """log of the prior-averaged likelihood (uniform prior on the box) by the trapezoid rule."""
    x = np.linspace(bounds[0, 0], bounds[1, 0], n)
    y = np.linspace(bounds[0, 1], bounds[1, 1], n)
    L = np.exp(-((1 - x[:, None]) ** 2 + 100 * (y[None, :] - x[:, None] ** 2) ** 2))
    integral = np.trapezoid(np.trapezoid(L, y, axis=1), x)
    return float(np.log(integral / ((x[-1] - x[0]) * (y[-1] - y[0]))))


if __name__ == "__main__":
    bounds = np.array([[-1, 4], [-1, 7]]).T
    t0 = time.time()
    with tempfile.TemporaryDirectory() as out:
        bobe = BOBE(loglikelihood=loglike, param_list=["x1", "x2"], param_bounds=bounds, param_labels=["x_1", "x_2"],
                    likelihood_name="Rosenbrock", verbosity="WARNING", n_sobol_init=8, optimizer="scipy",
                    use_clf=False, seed=42, save_dir=out, save=True)
        max_evals = int(os.environ.get("MAX_EVALS", 250))
        res = bobe.run(acq="wipstd", min_evals=25, max_evals=max_evals, max_gp_size=max_evals, logz_threshold=5e-2,
                       do_final_ns=True, fit_n_points=1, batch_size=1, ns_n_points=2, num_hmc_warmup=256,
                       num_hmc_samples=2048, mc_points_size=128, num_chains=4, convergence_n_iters=2)
        saved = os.path.exists(os.path.join(out, "Rosenbrock_gp.npz"))
    lz = res["logz"]
    print("rosenbrock-2d: %d evals in %.1fs, %s; checkpoint written: %s" % (res["gp"].npoints, time.time() - t0,
                                                                          res["termination_reason"], saved))
    print("logZ surrogate: mean %.3f [%.3f, %.3f]; direct quadrature: %.3f" % (lz["mean"], lz["lower"], lz["upper"],
                                                                              true_logz(bounds)))
    print("best logL %.4f at %s" % (res["best_val"], np.round(res["best_pt"], 4)))
    print("timing:", {k: round(v, 2) for k, v in res["timing"].items()})
