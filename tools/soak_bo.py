"""Soak run (GPU box): a BO loop that grows the GP well past the launch-bound sizes, re-capturing the evaluation graphs at
every new N, with the device-memory footprint sampled along the way."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402

D = 6


def loglike(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


free0 = torch.cuda.mem_get_info()[0]
bounds = np.array([[-2.0, 2.0]] * D).T
bobe = BOBE(loglike, [f"x{i}" for i in range(D)], bounds, n_sobol_init=64, seed=3, verbosity="WARNING")
t0 = time.time()
for target in (200, 400, 700, 1000):
    res = bobe.run(acq="wipstd", max_evals=target, max_gp_size=target, fit_n_points=10, batch_size=4, mc_points_size=128,
                   num_mc_samples=1024, mc_points_method="uniform")
    free = torch.cuda.mem_get_info()[0]
    print(f"N={res['gp'].npoints:5d}  t={time.time() - t0:7.1f}s  best={res['best_val']:.3f}  device memory in use by the "
          f"process: {(free0 - free) / 2**20:.0f} MiB  timing={ {k: round(v, 1) for k, v in res['timing'].items()} }", flush=True)
