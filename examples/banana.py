"""BASELINE.json config 1 — 2-D curved ("banana") likelihood, reference examples/Banana.py:14-18, 27-31, 54-68,
end to end on the GPU GP with WIPStd and uniform (Sobol) integration points."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402


def loglike(X):
    return -0.25 * (5 * (0.2 - X[0])) ** 2 - (20 * (X[1] / 4 - X[0] ** 4)) ** 2


if __name__ == "__main__":
    bounds = np.array([[-1, 1], [-1, 2]]).T
    t0 = time.time()
    bobe = BOBE(loglike, ["x1", "x2"], bounds, n_sobol_init=8, seed=42)
    res = bobe.run(acq="wipstd", max_evals=int(os.environ.get("MAX_EVALS", 60)), fit_n_points=1, batch_size=1,
                   mc_points_size=512)
    print("banana: %d evals in %.1fs, best logL %.4f at %s" % (res["n_evals"], time.time() - t0, res["best_val"], res["best_x"]))
    print("timing:", {k: round(v, 2) for k, v in res["timing"].items()})
