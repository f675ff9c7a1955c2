"""Config-5 probe: 10-D Rosenbrock, convergence trace of the BO run + nested sampling of the TRUE likelihood."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import samplers  # noqa: E402
from bobe_amd.bo import BOBE  # noqa: E402

D = 10
LO, HI = -2.0, 2.0


def rosen10(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


class TrueSurface:
    """duck-typed 'GP' whose mean is the true log-likelihood on the unit cube (variance zero)"""
    ndim = D

    def predict_mean_batched(self, u):
        x = LO + np.atleast_2d(u) * (HI - LO)
        return -np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1.0 - x[:, :-1]) ** 2, axis=1) / 20.0

    def predict_var_batched(self, u):
        return np.full(np.atleast_2d(u).shape[0], 1e-12)


if "truth" in sys.argv:
    for seed in (0, 1):
        t0 = time.time()
        _, lz, ok = samplers.nested_sampling(TrueSurface(), ndim=D, mode="convergence", rng=np.random.default_rng(seed), nlive=2000)
        print(f"true-likelihood NS seed {seed}: logZ {lz['mean']:.3f} +- {lz['dlogz_sampler']:.3f} ({lz['ncall']} calls, {time.time() - t0:.1f}s)", flush=True)
if "bo" in sys.argv:
    orig = samplers.nested_sampling

    def traced(gp, *a, **k):
        out = orig(gp, *a, **k)
        lz = out[1]
        print(f"   NS at N={gp.npoints}: mean {lz['mean']:.3f} upper {lz['upper']:.3f} lower {lz['lower']:.3f} std {lz['std']:.3f}", flush=True)
        return out
    samplers.nested_sampling = traced
    t0 = time.time()
    b = BOBE(rosen10, [f"x{i}" for i in range(D)], np.array([[LO, HI]] * D).T, n_sobol_init=64, seed=7)
    r = b.run(acq="wipstd", min_evals=150, max_evals=int(os.environ.get("MAX_EVALS", 1200)), max_gp_size=1500,
              logz_threshold=float(os.environ.get("THR", 1.0)), fit_n_points=10, ns_n_points=10, batch_size=5,
              mc_points_size=256, num_hmc_warmup=256, num_hmc_samples=512, do_final_ns=True)
    print(f"rosen10: {r['termination_reason']} after {r['n_evals']} evals in {time.time() - t0:.1f}s, logZ {r['logz']}; acq tail "
          f"{np.round(r['acq_history'][-5:], 3)}; timing { {k: round(v, 1) for k, v in r['timing'].items()} }", flush=True)
