"""BO-loop counterpart (SURVEY.md 8f "next" row 1) — a compact driver around the GPU GP.

Mirrors the parts of ``BOBE/bo.py`` that *call* the hot path: Sobol initialisation in the unit cube
(bo.py:505-537, utils/core.py:181-193), the initial multi-restart fit (bo.py:611 -> pool.py:268-293), the
WIPV / WIPStd / EI iteration (bo.py:1174-1224, 1226-1390: mc points -> get_next_batch -> evaluate ->
update_gp) and the refit policy of ``update_gp`` (bo.py:620-668, strict ``<`` size classes included).

Not reproduced: the MPI pool itself (its restart sharding is, over torch.distributed: ``gp_fit``),
NUTS.  The logZ convergence test (bo.py:886-891)
runs on ``bobe_amd.samplers.nested_sampling`` (batched on the GPU GP) instead of dynesty; the loop also stops
on ``max_evals``, ``max_gp_size`` or an acquisition-value threshold.  Integration points come from the
reference's ``'uniform'`` (scrambled Sobol, acquisition.py:476-479) or ``'NS'`` (acquisition.py:473-475) method.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
from scipy.stats import qmc

from .acquisition import EI, LogEI, WIPStd, WIPV, get_mc_samples
from .dist_sweep import dist_info, merge_best_fit, shard_bounds
from .gp import GP
from .utils import get_logger, scale_from_unit, scale_to_unit

log = get_logger("bo")

_ACQ = {"wipv": WIPV, "wipstd": WIPStd, "ei": EI, "logei": LogEI}


def gp_fit(gp: GP, maxiters: int = 1000, n_restarts: int = 8, rng: Optional[np.random.Generator] = None,
           group=None) -> dict:
    """``MPI_Pool.gp_fit`` (pool.py:268-328): x0 row 0 = log(current hp), further rows uniform in the log-bounds;
    fit; adopt the best hyper-parameters (refactors on the GPU).

    With an initialised ``torch.distributed`` group of G > 1 ranks (one process per GPU, every rank running the
    same loop with the same seed) the restarts are split over the ranks the way the reference's MPI pool splits
    them (``np.array_split``, pool.py:298-326): each rank runs its chunk — concurrently, on the evaluation slots of
    its own GPU — then one all-gather of (mll, theta) and max-by-mll; every rank adopts the same theta."""
    rng = np.random.default_rng() if rng is None else rng
    n_params = gp.hyperparam_bounds.shape[1]
    init = np.log(gp.get_hyperparams())
    if n_restarts > 1:
        x0 = np.vstack([init, rng.uniform(gp.hyperparam_bounds[0], gp.hyperparam_bounds[1],
                                          size=(n_restarts - 1, n_params))])
    else:
        x0 = np.atleast_2d(init)
    world, rank, coll_dev = dist_info(group, gp.device)
    if world > 1 and x0.shape[0] > 1:
        lo, hi = shard_bounds(x0.shape[0], world, rank)
        res = gp.fit(x0=x0[lo:hi], maxiter=maxiters) if hi > lo else {"mll": -np.inf, "params": init}
        mll, params = merge_best_fit(res["mll"], res["params"], group=group, device=coll_dev)
        res = {"mll": mll, "params": params}
    else:
        res = gp.fit(x0=x0, maxiter=maxiters)
    gp.update_hyperparams(res["params"])
    return res


class BOBE:
    """``BOBE(loglikelihood, param_list, param_bounds, ...).run(acq=...)`` with the GP on a MI355X."""

    def __init__(self, loglikelihood: Callable[[np.ndarray], float], param_list: Sequence[str] = None,
                 param_bounds: np.ndarray = None, param_labels=None, likelihood_name: Optional[str] = None,
                 confidence_for_unbounded=0.9999995, gp_kwargs: Optional[dict] = None, n_cobaya_init: int = 4,
                 n_sobol_init: int = 16, init_train_x=None, init_train_y=None, resume: bool = False, resume_file=None,
                 save_dir: str = ".", save: bool = False, save_step: int = 5, optimizer: str = "scipy",
                 acq: str = "WIPV", use_clf: bool = False, clf_type: str = "svm", clf_nsigma_threshold: float = 20,
                 clf_use_size: int = 10, clf_update_step: int = 1, minus_inf: float = -1e5,
                 seed: Optional[int] = None, verbosity: str = "INFO", device: int = 0):
        """Keywords of the reference constructor (bo.py:69-96) plus ``device``.  ``loglikelihood`` must be a callable
        on physical parameters (Cobaya likelihoods and ``resume`` belong to the parts that are not built);
        ``save`` writes ``<save_dir>/<likelihood_name>_gp.npz`` every ``save_step`` iterations (bo.py:239);
        ``use_clf`` selects ``GPwithClassifier`` (SVM) with the thresholds derived from ``clf_nsigma_threshold``."""
        import logging
        if not callable(loglikelihood):
            raise NotImplementedError("only a callable log-likelihood is supported (Cobaya adaptors are out of scope)")
        if resume:
            raise NotImplementedError("resume is not built; reload a saved GP with GP.load and pass init_train_x/y")
        if param_list is None or param_bounds is None:
            raise ValueError("param_list and param_bounds are required with a callable log-likelihood")
        logging.getLogger("bobe_amd").setLevel(getattr(logging, str(verbosity).upper(), logging.INFO))
        self.loglikelihood = loglikelihood
        self.param_list = list(param_list)
        self.param_labels = list(param_labels) if param_labels is not None else list(param_list)
        self.likelihood_name = likelihood_name or "likelihood"
        self.param_bounds = np.asarray(param_bounds, dtype=np.float64)      # (2, ndim), like the reference
        self.ndim = len(self.param_list)
        self.np_rng = np.random.default_rng(seed)
        self.minus_inf = float(minus_inf)
        self.device = device
        self.save, self.save_dir, self.save_step = bool(save), save_dir, max(1, int(save_step))
        self.default_acq = acq
        self.timing: Dict[str, float] = {"GP Training": 0.0, "Acquisition Optimization": 0.0,
                                         "True Objective Evaluations": 0.0}
        self.n_points_since_last_fit = 0
        # Sobol initial design (bo.py:521-529), optionally after user-supplied points (bo.py:505-519)
        n_sobol = max(2, n_sobol_init)
        sobol = qmc.Sobol(d=self.ndim, scramble=True, seed=self.np_rng).random(n_sobol)
        pts = scale_from_unit(sobol, self.param_bounds)
        vals = self._evaluate(pts)
        if init_train_x is not None and init_train_y is not None:
            pts = np.vstack([np.atleast_2d(np.asarray(init_train_x, dtype=np.float64)), pts])
            vals = np.vstack([np.asarray(init_train_y, dtype=np.float64).reshape(-1, 1), vals])
        kw = dict(noise=1e-8, kernel="rbf", lengthscale_bounds=[0.01, 10], kernel_variance_bounds=[1e-4, 1e8],
                  optimizer=optimizer)
        kw.update(gp_kwargs or {})
        t0 = time.time()
        x_u = scale_to_unit(pts, self.param_bounds)
        if use_clf:
            from .clf_gp import GPwithClassifier
            from .utils import get_threshold_for_nsigma
            clf_threshold = max(75.0, get_threshold_for_nsigma(clf_nsigma_threshold, self.ndim))     # bo.py:593
            self.gp = GPwithClassifier(x_u, vals, clf_type=clf_type, clf_use_size=clf_use_size,
                                       clf_update_step=clf_update_step, probability_threshold=0.5,
                                       minus_inf=self.minus_inf, clf_threshold=clf_threshold,
                                       gp_threshold=2 * clf_threshold, param_names=self.param_list,
                                       device=device, **kw)
        else:
            self.gp = GP(x_u, vals, param_names=self.param_list, device=device, **kw)
        gp_fit(self.gp, n_restarts=4, maxiters=500, rng=self.np_rng)        # bo.py:611
        self.timing["GP Training"] += time.time() - t0

    def _evaluate(self, pts: np.ndarray) -> np.ndarray:
        """Safe likelihood wrapper (likelihood.py:69-91): NaN / exceptions / -inf -> minus_inf."""
        t0 = time.time()
        out = np.empty((pts.shape[0], 1))
        for i, p in enumerate(pts):
            try:
                v = float(self.loglikelihood(np.array(p)))
            except Exception:
                v = self.minus_inf
            out[i, 0] = v if np.isfinite(v) and v > self.minus_inf else self.minus_inf
        self.timing["True Objective Evaluations"] += time.time() - t0
        return out

    def update_gp(self, new_pts_u: np.ndarray, new_vals: np.ndarray, fit_n_points: int) -> None:
        """bo.py:620-668 — refit thresholds by training-set size (strict '<' as in the reference)."""
        t0 = time.time()
        self.n_points_since_last_fit += new_pts_u.shape[0]
        n = self.gp.train_x.shape[0]
        if n < 200:
            refit_threshold, maxiter, n_restarts = min(2, fit_n_points), 1000, 8
        elif 200 < n < 750:
            refit_threshold, n_restarts, maxiter = fit_n_points, 4, 500
        else:
            refit_threshold, n_restarts, maxiter = max(40, fit_n_points), 4, 200
        refit = self.n_points_since_last_fit >= refit_threshold
        self.gp.update(new_pts_u, new_vals)
        if getattr(self.gp, "not_pd", False):        # the new point made K numerically singular at the old
            refit = True                             # hyper-parameters (NaN factor, like XLA): refit now
        if refit:
            gp_fit(self.gp, n_restarts=n_restarts, maxiters=maxiter, rng=self.np_rng)
            self.n_points_since_last_fit = 0
        self.timing["GP Training"] += time.time() - t0

    def run(self, acq=None, min_evals: int = 0, max_evals: int = 250, max_gp_size: int = 1200,
            fit_n_points: int = 10, batch_size: int = 1, mc_points_size: int = 64, num_mc_samples: int = 1024,
            mc_points_method: str = "NUTS", logz_threshold: Optional[float] = None, ns_n_points: int = 10,
            convergence_n_iters: int = 1, do_final_ns: bool = False, acq_threshold: Optional[float] = None,
            zeta_ei: float = 0.01, verbose: bool = False, ei_goal: Optional[float] = None, num_hmc_warmup: int = 512,
            num_hmc_samples: int = 512, thinning: int = 4, num_chains: int = 4) -> dict:
        """BO loop.  With ``logz_threshold`` the run also stops once nested sampling on the surrogate gives
        (logZ_upper - logZ_lower)/2 < threshold ``convergence_n_iters`` times in a row (bo.py:886-891, 1283-1311);
        the check runs every ``ns_n_points`` new evaluations after ``min_evals``."""
        from .samplers import nested_sampling
        acq = acq if acq is not None else self.default_acq
        if isinstance(acq, (tuple, list)):                       # the reference accepts a tuple of stages: first one
            acq = acq[0]
        acq_fn = _ACQ[acq.lower()]()
        is_wip = acq.lower() in ("wipv", "wipstd")
        acq_hist: List[float] = []
        logz: Optional[dict] = None
        samples: dict = {}
        converged, n_ok, since_ns = False, 0, 0
        reason, it = None, 0
        self.timing.setdefault("Nested Sampling", 0.0)
        # the reference counts objective evaluations (bo.py:1198, 765-775): a proposal the GP rejects as a duplicate
        # still counts, so the loop ends by max_evals even when nothing new is accepted
        current_evals = self.gp.npoints
        while current_evals < max_evals and self.gp.npoints < max_gp_size:
            it += 1
            t0 = time.time()
            if is_wip:
                if mc_points_method == "NUTS":               # bo.py:1243-1250: HMC settings of run()
                    mc = get_mc_samples(self.gp, warmup_steps=num_hmc_warmup, num_samples=num_hmc_samples,
                                        thinning=thinning, method="NUTS", num_chains=num_chains, np_rng=self.np_rng)
                else:
                    mc = get_mc_samples(self.gp, num_samples=num_mc_samples, method=mc_points_method,
                                        np_rng=self.np_rng)
                samples = mc
                kwargs = {"mc_samples": mc, "mc_points_size": mc_points_size}
                new_u, vals = acq_fn.get_next_batch(self.gp, n_batch=batch_size, acq_kwargs=kwargs, n_restarts=1,
                                                    maxiter=100, early_stop_patience=10, rng=self.np_rng)  # bo.py:1274
            else:
                kwargs = {"zeta": zeta_ei, "best_y": float(np.max(self.gp.train_y))}
                new_u, vals = acq_fn.get_next_batch(self.gp, n_batch=1, acq_kwargs=kwargs, n_restarts=20,
                                                    maxiter=250, rng=self.np_rng)
            new_u = np.atleast_2d(new_u)
            self.timing["Acquisition Optimization"] += time.time() - t0
            acq_hist.append(float(np.mean(vals)))
            new_vals = self._evaluate(scale_from_unit(new_u, self.param_bounds))
            self.update_gp(new_u, new_vals, fit_n_points)
            if verbose:
                log.info(f"N={self.gp.npoints} acq={acq_hist[-1]:.3e}")
            if self.save and it % self.save_step == 0:
                import os
                self.gp.save(os.path.join(self.save_dir, f"{self.likelihood_name}_gp"))
            current_evals += new_u.shape[0]
            if acq_threshold is not None and is_wip and acq_hist[-1] <= acq_threshold:
                reason = "Acquisition threshold reached"
                break
            if ei_goal is not None and not is_wip and current_evals >= min_evals:          # bo.py:1208-1215
                goal_val = np.exp(acq_hist[-1]) if acq.lower() == "logei" else acq_hist[-1]
                if goal_val < ei_goal:
                    converged, reason = True, f"{acq_fn.name.upper()} goal reached"
                    break
            since_ns += new_u.shape[0]
            if logz_threshold is not None and current_evals >= min_evals and since_ns >= ns_n_points:
                t0 = time.time()
                _, logz, ok = nested_sampling(self.gp, mode="convergence", rng=self.np_rng)
                self.timing["Nested Sampling"] += time.time() - t0
                since_ns = 0
                delta = (logz["upper"] - logz["lower"]) / 2.0                    # bo.py:886-891
                n_ok = n_ok + 1 if (ok and delta < logz_threshold) else 0
                if n_ok >= convergence_n_iters:
                    converged, reason = True, "LogZ converged"
                    break
        if do_final_ns or (logz_threshold is not None and logz is None):
            t0 = time.time()
            ns_samples, logz, _ = nested_sampling(self.gp, mode="convergence", rng=self.np_rng)
            samples = ns_samples if isinstance(ns_samples, dict) else samples
            self.timing["Nested Sampling"] += time.time() - t0
        if reason is None:                                       # bo.py:769-774
            reason = "Maximum GP size reached" if self.gp.npoints >= max_gp_size and max_gp_size <= max_evals \
                else "Maximum evaluations reached"
        if self.save:
            import os
            self.gp.save(os.path.join(self.save_dir, f"{self.likelihood_name}_gp"))
        y = self.gp.train_y * self.gp.y_std + self.gp.y_mean
        ibest = int(np.argmax(y))
        best_x = scale_from_unit(self.gp.train_x[ibest], self.param_bounds)
        manager = {"likelihood_name": self.likelihood_name, "param_list": self.param_list,
                   "param_labels": self.param_labels, "acquisition_history": list(acq_hist),
                   "timing": dict(self.timing), "converged": converged, "termination_reason": reason,
                   "gp_training_set_size": int(self.gp.npoints)}
        # keys of the reference's results dict (bo.py:827-836): EI / LogEI runs carry empty 'samples' and 'logz'
        return {"gp": self.gp, "likelihood": self.loglikelihood, "results_manager": manager,
                "best_val": float(y[ibest, 0]), "best_pt": best_x,
                "termination_reason": reason,
                "samples": (samples if is_wip else {}),
                "best_x": best_x,
                "n_evals": int(self.gp.npoints), "acq_history": acq_hist, "timing": dict(self.timing),
                "lengthscales": np.array(self.gp.lengthscales), "kernel_variance": float(self.gp.kernel_variance),
                "logz": (logz if logz is not None else {}), "converged": converged}
