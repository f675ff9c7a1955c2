"""BASELINE config 5: the 10-D Rosenbrock BO loop to the reference's stopping rule (bo.py:886-934), with a trace.

  python tools/config5_run.py [key=value ...]
     max_evals=3200 max_gp=4096 thr=1.0 n_iters=2 batch=5 ns_every=50 min_evals=400 clf=0 seed=7 mc=256 fit_every=10
     dim=10 (the same function in another dimension)  truth=1 (only the nested sampling of the true likelihood)
     steptime=1 (only: microseconds per leapfrog / random-walk step of the sampler kernels at several (N, d))
Prints every nested-sampling result (N, logZ mean / upper / lower, half-width), the state at the checkpoints
N = 600, 1200, 2400, 4096 (phase timers so far) and the final line.  `truth` = nested sampling of the TRUE likelihood
with the same sampler (2000 live points): logZ = -15.6 +- 0.1 (profiles/r03_configs_1_and_5_runs.txt)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import samplers  # noqa: E402
from bobe_amd.bo import BOBE  # noqa: E402

LO, HI = -2.0, 2.0
opt = dict(max_evals=3200, max_gp=4096, thr=1.0, n_iters=2, batch=5, ns_every=50, min_evals=400, clf=0, seed=7, mc=256,
           fit_every=10, sobol=64, warmup=256, hmc=512, dim=10, truth=0, steptime=0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    opt[k] = type(opt[k])(float(v)) if k in opt else v
D = opt["dim"]


def rosen10(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


if opt["steptime"]:
    from bobe_amd import GP
    for N, d in ((64, 10), (600, 10), (1024, 10), (2048, 10), (3200, 16), (4096, 16), (4096, 8), (4096, 20), (1000, 20)):
        rng = np.random.default_rng(0)
        X = rng.uniform(size=(N, d))
        gp = GP(X, -20.0 * np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.8), kernel_variance=2.0)
        P, niter = 64, 64
        x0 = rng.uniform(0.3, 0.7, size=(P, d))
        m, _, dm, _ = gp.predict_grad(x0, mean_only=True)
        mean = m * gp.y_std + gp.y_mean
        g = dm * gp.y_std * (x0 * (1 - x0)) + (1 - 2 * x0)
        lp = mean + np.sum(np.log(x0) + np.log1p(-x0), axis=1)
        state = np.ascontiguousarray(np.concatenate([np.log(x0) - np.log1p(-x0), g, x0, lp[:, None], mean[:, None]], axis=1))
        adapt = np.tile(np.array([0.05, 0.0, 0.0, 0.0, 0.0]), (P, 1))
        gp.hmc_run(state.copy(), adapt.copy(), np.ones(d), 1, 0, 4, False, 1.0)
        t0 = time.perf_counter()
        gp.hmc_run(state.copy(), adapt.copy(), np.ones(d), 1, 0, niter, False, 1.0)
        dt = time.perf_counter() - t0
        l0 = gp.predict_mean_batched(x0)
        t1 = time.perf_counter()
        gp.rwalk(x0, l0, 0.02 * np.eye(d), -1e30, 512, seed=3)
        dr = time.perf_counter() - t1
        print(f"N={N:5d} d={d:2d}: hmc {dt / niter / 8 * 1e6:7.2f} us per leapfrog step (64 chains, 8 steps per trajectory on average), "
              f"random walk {dr / 512 * 1e6:7.2f} us per step (64 walkers)", flush=True)
    sys.exit(0)

if opt["truth"]:
    class Truth:                                            # the sampler's view of a surrogate, on the true likelihood
        ndim = D

        def predict_mean_batched(self, u):
            x = LO + (HI - LO) * np.atleast_2d(u)
            return -np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1.0 - x[:, :-1]) ** 2, axis=1) / 20.0

        def predict_var_batched(self, u):                   # (no surrogate error: upper = lower = mean)
            return np.zeros(len(np.atleast_2d(u)))

    for sd in (0, 1):
        t0 = time.time()
        _, lz, ok = samplers.nested_sampling(Truth(), nlive=2000, rng=np.random.default_rng(sd), device_walks=False)
        print(f"truth dim={D} seed {sd}: logZ {lz['mean']:.3f} +- {lz.get('dlogz_sampler', float('nan')):.3f} ncall {lz.get('ncall')} "
              f"ok {ok} ({time.time() - t0:.0f}s)", flush=True)
    sys.exit(0)

checkpoints = [600, 1200, 2400, 4096]
t_start = time.time()
bobe = None
orig_ns = samplers.nested_sampling


def traced(gp, *a, **k):
    t0 = time.time()
    out = orig_ns(gp, *a, **k)
    lz = out[1]
    print(f"   NS at N={gp.npoints:5d} t={time.time() - t_start:7.1f}s: mean {lz['mean']:8.3f} upper {lz['upper']:8.3f} lower "
          f"{lz['lower']:8.3f} half-width {(lz['upper'] - lz['lower']) / 2:7.3f} ncall {lz.get('ncall')} "
          f"({time.time() - t0:.1f}s){' TRUNCATED' if lz.get('truncated') else ''}", flush=True)
    return out


samplers.nested_sampling = traced
orig_update = BOBE.update_gp


def traced_update(self, new_u, new_vals, step=0, verbose=True):
    before = self.gp.npoints
    orig_update(self, new_u, new_vals, step=step, verbose=verbose)
    for c in checkpoints:
        if before < c <= self.gp.npoints:
            print(f"== checkpoint N={self.gp.npoints} at {time.time() - t_start:.1f}s: timing "
                  f"{ {k: round(v, 1) for k, v in self.timing.items()} } ls {np.round(self.gp.lengthscales, 3)} kvar "
                  f"{self.gp.kernel_variance:.3g} y_std {self.gp.y_std:.4g}", flush=True)


BOBE.update_gp = traced_update
bobe = BOBE(rosen10, [f"x{i}" for i in range(D)], np.array([[LO, HI]] * D).T, n_sobol_init=opt["sobol"], seed=opt["seed"],
            use_clf=bool(opt["clf"]), minus_inf=-1e10, save=False)
res = bobe.run(acq="wipstd", min_evals=opt["min_evals"], max_evals=opt["max_evals"], max_gp_size=opt["max_gp"],
               logz_threshold=opt["thr"], convergence_n_iters=opt["n_iters"], fit_n_points=opt["fit_every"],
               ns_n_points=opt["ns_every"], batch_size=opt["batch"], mc_points_size=opt["mc"], num_hmc_warmup=opt["warmup"],
               num_hmc_samples=opt["hmc"], do_final_ns=True)
lz = res["logz"]
print(f"rosen10 {opt}: {res['termination_reason']} after {res['n_evals']} GP points in {time.time() - t_start:.1f}s; logZ mean "
      f"{lz.get('mean', float('nan')):.3f} upper {lz.get('upper', float('nan')):.3f} lower {lz.get('lower', float('nan')):.3f} "
      f"half-width {(lz.get('upper', 0) - lz.get('lower', 0)) / 2:.3f}; best logL {res['best_val']:.3f}; timing "
      f"{ {k: round(v, 1) for k, v in res['timing'].items()} }", flush=True)
