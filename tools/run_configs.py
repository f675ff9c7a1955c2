"""Scratch driver for BASELINE configs 1 and 5 on the GPU box (numbers quoted in DESIGN.md come from here)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.bo import BOBE  # noqa: E402

HELD = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_held.json")))


def banana(x):
    return -0.25 * (5 * (0.2 - x[0])) ** 2 - (20 * (x[1] / 4 - x[0] ** 4)) ** 2


def himmelblau(x):
    return -0.5 * (0.1 * (x[0] + x[1] ** 2 - 7) ** 2 + (x[0] ** 2 + x[1] - 11) ** 2)


def rosen10(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


if os.environ.get("HMC_STEPWISE"):                      # A/B: one surrogate call per leapfrog step instead of one per trajectory
    from bobe_amd.gp import GP
    del GP.hmc_leapfrog
SEED = int(os.environ.get("SEED", 7))
which = sys.argv[1:] or ["banana", "himmelblau", "rosen10"]
if "banana" in which:
    h = HELD["notebook_banana"]
    for seed in (h["seed"], 1, 2):
        t0 = time.time()
        b = BOBE(banana, h["param_list"], np.array(h["param_bounds"]).T, n_sobol_init=2, seed=seed, save=False)
        hp0 = b.gp.hyperparams_dict()
        r = b.run(**h["run"])
        print(f"banana seed {seed}: first fit {hp0} | {r['termination_reason']} after {r['n_evals']} evals, logZ "
              f"{r['logz'].get('mean', float('nan')):.4f} +- {(r['logz'].get('upper', 0) - r['logz'].get('lower', 0)) / 2:.4f}"
              f" in {time.time() - t0:.1f}s; samples x range {r['samples']['x'].min(0)} .. {r['samples']['x'].max(0)}", flush=True)
if "himmelblau" in which:
    t0 = time.time()
    b = BOBE(himmelblau, ["x1", "x2"], np.array([[-4, 4], [-4, 4]]).T, n_sobol_init=8, seed=42, save=False)
    r = b.run(acq="wipstd", min_evals=25, max_evals=250, logz_threshold=0.01, fit_n_points=4, batch_size=2, ns_n_points=4,
              num_hmc_warmup=256, num_hmc_samples=512, mc_points_size=128, convergence_n_iters=1)
    print(f"himmelblau (detailed_usage.rst settings): {r['termination_reason']} after {r['n_evals']} evals, logZ "
          f"{r['logz'].get('mean', float('nan')):.4f} +- {(r['logz'].get('upper', 0) - r['logz'].get('lower', 0)) / 2:.4f} in {time.time() - t0:.1f}s",
          flush=True)
if "rosen10" in which:
    t0 = time.time()
    D = 10
    b = BOBE(rosen10, [f"x{i}" for i in range(D)], np.array([[-2.0, 2.0]] * D).T, n_sobol_init=64, seed=SEED, save=False)
    r = b.run(acq="wipstd", min_evals=150, max_evals=int(os.environ.get("MAX_EVALS", 600)), logz_threshold=0.5, fit_n_points=10,
              ns_n_points=10, batch_size=5, mc_points_size=256, num_hmc_warmup=256, num_hmc_samples=512, do_final_ns=True,
              verbose=True)
    print(f"rosen10: {r['termination_reason']} after {r['n_evals']} evals, logZ {r['logz']} in {time.time() - t0:.1f}s; timing "
          f"{ {k: round(v, 1) for k, v in r['timing'].items()} }", flush=True)
