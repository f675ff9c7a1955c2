"""Seeded fixtures for the callers either side of the hot path (last bullet): a short WIPStd BO run of the
ORACLE (oracle/bobe_oracle_loop.py) on the 2-D Himmelblau likelihood of the reference's tests/test_bo_2d.py — the
kriging-believer batches it chose, the integration points it drew, the refit schedule and the fitted
hyper-parameters, step by step.  tests/test_gpu_loop_parity.py replays every step on the GPU from the recorded state
("teacher forcing": one step never inherits another's rounding) and must choose the same points.

    python tests/golden/make_loop_golden.py     ->  tests/golden/loop_himmelblau.npz

These are oracle outputs (parity unpinned, like the other .npz files here), not reference outputs.
"""
import os
import sys

import numpy as np
from scipy.stats import qmc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bobe_oracle as O  # noqa: E402
from oracle import bobe_oracle_loop as OL  # noqa: E402

BOUNDS = np.array([[-4.0, 4.0], [-4.0, 4.0]]).T
N_ITERS, N_BATCH, MC_SIZE, NUM_MC, FIT_N_POINTS = 5, (1, 1, 3, 3, 3), 48, 256, 4     # batch size per step


def himmelblau(x):
    return -0.5 * (0.1 * (x[0] + x[1] ** 2 - 7) ** 2 + (x[0] ** 2 + x[1] - 11) ** 2)


def to_phys(u):
    return BOUNDS[0] + u * (BOUNDS[1] - BOUNDS[0])


def main():
    rng = np.random.default_rng(123)
    unit = qmc.Sobol(d=2, scramble=True, seed=rng).random(8)
    vals = np.array([himmelblau(p) for p in to_phys(unit)]).reshape(-1, 1)
    gp = O.OracleGP(unit, vals)
    x0 = O.restart_points(np.log(gp.get_hyperparams()), gp.hyperparam_bounds, 4, rng)
    gp.update_hyperparams(gp.fit(x0=x0, maxiter=500)["params"])
    out = {"bounds": BOUNDS, "n_iters": N_ITERS, "n_batch": np.array(N_BATCH), "mc_points_size": MC_SIZE,
           "num_mc_samples": NUM_MC, "fit_n_points": FIT_N_POINTS}
    n_since = 0
    for it in range(N_ITERS):
        # state entering the step
        out[f"s{it}_train_x"] = np.array(gp.train_x)
        out[f"s{it}_train_y"] = np.array(gp.train_y * gp.y_std + gp.y_mean)
        out[f"s{it}_lengthscales"] = np.array(gp.lengthscales)
        out[f"s{it}_kernel_variance"] = float(gp.kernel_variance)
        out[f"s{it}_n_since"] = n_since
        mc_x = qmc.Sobol(d=2, scramble=True, seed=np.random.default_rng(3000 + it)).random(NUM_MC)   # 'uniform' (acquisition.py:476-479)
        out[f"s{it}_mc_samples"] = mc_x
        xs, acq, infos = OL.get_next_batch(gp, "wipstd", mc_x, MC_SIZE, N_BATCH[it], np.random.default_rng(1000 + it))
        out[f"s{it}_batch_x"] = xs
        out[f"s{it}_batch_val"] = acq
        out[f"s{it}_mc_points"] = np.array([i["mc_points"] for i in infos])
        out[f"s{it}_sweep_index"] = np.array([i["sweep_index"] for i in infos])
        out[f"s{it}_sweep_value"] = np.array([i["sweep_value"] for i in infos])
        new_y = np.array([himmelblau(p) for p in to_phys(xs)]).reshape(-1, 1)
        out[f"s{it}_new_y"] = new_y
        n_since, refit, n_restarts, maxiter = OL.update_gp(gp, xs, new_y, n_since, FIT_N_POINTS, np.random.default_rng(2000 + it))
        out[f"s{it}_refit"] = bool(refit)
        out[f"s{it}_n_restarts"] = n_restarts
        out[f"s{it}_maxiter"] = maxiter
        out[f"s{it}_after_lengthscales"] = np.array(gp.lengthscales)
        out[f"s{it}_after_kernel_variance"] = float(gp.kernel_variance)
        out[f"s{it}_after_neg_mll"] = float(gp.neg_mll(np.log(gp.get_hyperparams())))
        print(f"step {it}: N={gp.train_x.shape[0]} batch={np.round(xs, 4).tolist()} refit={refit} ls={np.round(gp.lengthscales, 4)}")
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "loop_himmelblau.npz"), **out)


if __name__ == "__main__":
    main()
