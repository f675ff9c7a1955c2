"""BASELINE.json config 5 — 10-D Rosenbrock log-likelihood, full BO loop on the GPU GP.
(The reference's examples/Rosenbrock.py is 2-D, bounds [-1,4]x[-1,7]; the 10-D variant is built here.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402

D = 10


def loglike(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


if __name__ == "__main__":
    bounds = np.array([[-2.0, 2.0]] * D).T
    t0 = time.time()
    bobe = BOBE(loglike, [f"x{i}" for i in range(D)], bounds, n_sobol_init=64, seed=7)
    res = bobe.run(acq="wipstd", max_evals=int(os.environ.get("MAX_EVALS", 300)), fit_n_points=10, batch_size=4,
                   mc_points_size=256, num_mc_samples=4096)
    print("rosenbrock-10d: %d evals in %.1fs, best logL %.3f" % (res["n_evals"], time.time() - t0, res["best_val"]))
    print("timing:", {k: round(v, 2) for k, v in res["timing"].items()})
