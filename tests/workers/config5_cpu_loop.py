"""BASELINE config 5 on the HOST cores: the 10-D Rosenbrock BO loop with the CPU oracle as the surrogate, for a wall-clock
budget.  Test infrastructure (it imports oracle/): the CPU side of profiles/r04_config5.txt, never shipped or measured as
the product.

  python tests/workers/config5_cpu_loop.py [budget_s=600] [seed=7] [threads=16]

Same problem, seed, Sobol design and run settings as tools/config5_run.py (batch 5, 256 integration points, HMC 256 + 512,
nested sampling every 50 evaluations past 400, refit policy of bo.py:632-655).  Surrogate = oracle.OracleGP (NumPy / SciPy
-> LAPACK), acquisition = oracle.bobe_oracle_loop.get_next_batch (the rank-1 sweep, kriging believer with full
refactorisations as in the reference), samplers = the product's HOST-side sampler logic (bobe_amd/samplers.py is plain NumPy
around a duck-typed surrogate: nested sampling in batches, HMC stepped from the host), posterior-mean gradient in NumPy.
Left out, in the CPU's favour: the local refinement of every acquisition point for N <= 500 (jax.grad in the reference,
acquisition.py:403-412; the oracle's stand-in is a central difference of the literal fantasy variance, minutes per point).
Prints one line per iteration (N, seconds by phase) and a summary when the budget is spent."""
import os
import sys
import time

import numpy as np
from scipy.stats import qmc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
opt = dict(budget_s=600.0, seed=7, threads=16, batch=5, mc=256, ns_every=50, min_evals=400, thr=1.0, fit_every=10)
for a in sys.argv[1:]:
    k, v = a.split("=")
    opt[k] = type(opt[k])(float(v))
os.environ.setdefault("OMP_NUM_THREADS", str(opt["threads"]))
os.environ.setdefault("MKL_NUM_THREADS", str(opt["threads"]))
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(opt["threads"]))

from bobe_amd import samplers  # noqa: E402   (host-side sampler logic only: no library call is made)
from oracle import bobe_oracle as O  # noqa: E402
from oracle import bobe_oracle_loop as OL  # noqa: E402

D, LO, HI = 10, -2.0, 2.0


def rosen10(x):
    x = np.asarray(x)
    return -float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)) / 20.0


class CpuGP(O.OracleGP):
    """OracleGP + the standardised posterior-mean gradient the host-stepped HMC asks for (RBF: dk/dx = k (x_n - x) / ls^2)."""

    def predict_grad(self, x, mean_only=True):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        k = self._k12(x)                                        # (N, C)
        a = np.asarray(self.alphas).reshape(-1)
        mean = k.T @ a
        w = k * a[:, None]                                      # (N, C)
        dm = (w.T @ self.train_x - np.sum(w, axis=0)[:, None] * x) / (np.asarray(self.lengthscales) ** 2)[None, :]
        return mean, None, dm, None


t_start = time.time()
rng = np.random.default_rng(opt["seed"])
unit = qmc.Sobol(d=D, scramble=True, seed=rng).random(64)
vals = np.array([rosen10(LO + u * (HI - LO)) for u in unit]).reshape(-1, 1)
gp = CpuGP(unit, vals)
x0 = O.restart_points(np.log(gp.get_hyperparams()), gp.hyperparam_bounds, 4, rng)
gp.update_hyperparams(gp.fit(x0=x0, maxiter=500)["params"])
timing = {"fit": time.time() - t_start, "acq": 0.0, "hmc": 0.0, "ns": 0.0}


def hmc():
    t0 = time.time()
    s = samplers.sample_GP_NUTS(gp, np_rng=rng, num_chains=4, warmup_steps=256, num_samples=512, thinning=4,
                                fused_trajectories=False)
    timing["hmc"] += time.time() - t0
    return s


mc = hmc()
n_since, n_since_ns, it, counter = 0, 0, 0, 0
current = gp.npoints
print(f"# CPU oracle loop, {opt['threads']} threads, budget {opt['budget_s']:.0f} s; initial fit + HMC: {time.time() - t_start:.1f} s", flush=True)
while time.time() - t_start < opt["budget_s"]:
    it += 1
    t_it = time.time()
    n_since_ns += opt["batch"]
    ns_flag = n_since_ns >= opt["ns_every"] and current >= opt["min_evals"]
    t0 = time.time()
    xs, acq, _ = OL.get_next_batch(gp, "wipstd", mc["x"], opt["mc"], opt["batch"], rng, maxiter=100, refine=False)
    timing["acq"] += time.time() - t0
    new_y = np.array([rosen10(LO + u * (HI - LO)) for u in xs]).reshape(-1, 1)
    current += opt["batch"]
    t0 = time.time()
    n_since, refit, _, _ = OL.update_gp(gp, xs, new_y, n_since, opt["fit_every"], rng)
    timing["fit"] += time.time() - t0
    note = ""
    if ns_flag and float(acq[-1]) <= opt["thr"]:
        t0 = time.time()
        smp, lz, ok = samplers.nested_sampling(gp, mode="convergence", dlogz=0.01, rng=rng)
        timing["ns"] += time.time() - t0
        n_since_ns = 0
        if ok:
            eq_x, eq_l = samplers.resample_equal(smp["x"], smp["logl"], smp["weights"], rng=rng)
            mc = {"x": eq_x}
            hw = (lz["upper"] - lz["lower"]) / 2
            counter = counter + 1 if hw < opt["thr"] else 0
            note = f" NS: logZ {lz['mean']:.3f} +- {hw:.3f}"
            if counter >= 2:
                print(f"it {it:4d} N={gp.npoints:5d}{note}: LogZ converged at {time.time() - t_start:.1f} s", flush=True)
                break
    else:
        mc = hmc()
    print(f"it {it:4d} N={gp.npoints:5d} iteration {time.time() - t_it:6.2f} s (refit {int(refit)}){note}  | so far {time.time() - t_start:7.1f} s: "
          f"{ {k: round(v, 1) for k, v in timing.items()} }", flush=True)
print(f"CPU oracle loop: N = {gp.npoints} after {time.time() - t_start:.1f} s and {it} iterations on {opt['threads']} threads; "
      f"timing { {k: round(v, 1) for k, v in timing.items()} }", flush=True)
