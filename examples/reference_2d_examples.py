"""The reference's four analytic 2-D examples (examples/Banana.py, Rosenbrock.py, Himmelblau.py, GaussianRing.py) with
their own likelihoods, bounds, constructor and run settings, on the GPU GP; each run's surrogate evidence is printed
next to a direct quadrature of the likelihood over the prior box.

    python examples/reference_2d_examples.py [banana|rosenbrock|himmelblau|ring ...]     (default: all four)
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402

COMMON = dict(acq="wipstd", min_evals=25, max_evals=250, max_gp_size=250, logz_threshold=5e-2, do_final_ns=True,
              num_chains=4, convergence_n_iters=2)

CASES = {
    # name: (log-likelihood on arrays, bounds (2, ndim), run settings that differ between the examples)
    "banana": (lambda x, y: -0.25 * (5 * (0.2 - x)) ** 2 - (20 * (y / 4 - x ** 4)) ** 2,          # Banana.py:14-18
               np.array([[-1, 1], [-1, 2]]).T,
               dict(fit_n_points=1, batch_size=1, ns_n_points=1, num_hmc_warmup=512, num_hmc_samples=2048,
                    mc_points_size=512)),                                                           # Banana.py:53-68
    "rosenbrock": (lambda x, y: -((1 - x) ** 2 + 100 * (y - x ** 2) ** 2),                        # Rosenbrock.py:14-16
                   np.array([[-1, 4], [-1, 7]]).T,
                   dict(fit_n_points=1, batch_size=1, ns_n_points=2, num_hmc_warmup=256, num_hmc_samples=2048,
                        mc_points_size=128)),                                                       # Rosenbrock.py:52-66
    "himmelblau": (lambda x, y: -0.5 * (0.1 * (x + y ** 2 - 7) ** 2 + (x ** 2 + y - 11) ** 2),    # Himmelblau.py:16-23
                   np.array([[-4, 4], [-4, 4]]).T,
                   dict(fit_n_points=2, batch_size=2, ns_n_points=2, num_hmc_warmup=512, num_hmc_samples=2048,
                        mc_points_size=512)),                                                       # Himmelblau.py:52-66
    "ring": (lambda x, y: -0.5 * ((np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2) - 0.2) / 0.02) ** 2,  # GaussianRing.py:16-22
             np.array([[0, 1], [0, 1]]).T,
             dict(fit_n_points=2, batch_size=2, ns_n_points=4, num_hmc_warmup=512, num_hmc_samples=2048,
                  mc_points_size=512)),                                                             # GaussianRing.py:54-68
}


def true_logz(fn, bounds, n=3001):
    """log of the prior-averaged likelihood (uniform prior on the box), trapezoid rule."""
    x = np.linspace(bounds[0, 0], bounds[1, 0], n)
    y = np.linspace(bounds[0, 1], bounds[1, 1], n)
    L = np.exp(fn(x[:, None], y[None, :]))
    return float(np.log(np.trapezoid(np.trapezoid(L, y, axis=1), x) / ((x[-1] - x[0]) * (y[-1] - y[0]))))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    max_evals = int(os.environ.get("MAX_EVALS", COMMON["max_evals"]))
    for name in names:
        fn, bounds, extra = CASES[name]
        t0 = time.time()
        with tempfile.TemporaryDirectory() as out:
            bobe = BOBE(loglikelihood=lambda p, fn=fn: float(fn(p[0], p[1])), param_list=["x1", "x2"],
                        param_bounds=bounds, param_labels=["x_1", "x_2"], likelihood_name=name, verbosity="WARNING",
                        n_sobol_init=8, optimizer="scipy", use_clf=False, seed=42, save_dir=out, save=True)
            kw = dict(COMMON, **extra)
            kw.update(max_evals=max_evals, max_gp_size=max_evals)
            res = bobe.run(**kw)
        lz = res["logz"]
        print("%-11s %3d evaluations, %5.1f s, %-28s logZ %.3f [%.3f, %.3f]  quadrature %.3f" % (
            name, res["gp"].npoints, time.time() - t0, res["termination_reason"] + ";", lz.get("mean", np.nan),
            lz.get("lower", np.nan), lz.get("upper", np.nan), true_logz(fn, bounds)), flush=True)
