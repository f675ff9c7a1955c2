"""cProfile of a 10-D Rosenbrock BO run (config 5, shortened): where the host time of the loop goes."""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE


def rosen10(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


D = 10
b = BOBE(rosen10, [f"x{i}" for i in range(D)], np.array([[-2.0, 2.0]] * D).T, n_sobol_init=64, seed=7)
pr = cProfile.Profile()
pr.enable()
r = b.run(acq="wipstd", min_evals=150, max_evals=int(os.environ.get("MAX_EVALS", 350)), logz_threshold=0.5, fit_n_points=10,
          ns_n_points=10, batch_size=5, mc_points_size=256, num_hmc_warmup=256, num_hmc_samples=512, do_final_ns=False)
pr.disable()
print(r["termination_reason"], r["n_evals"], {k: round(v, 2) for k, v in r["timing"].items()})
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
