"""BO-loop counterpart (SURVEY.md 8f "next" row 1) — a compact driver around the GPU GP.

Mirrors the parts of ``BOBE/bo.py`` that *call* the hot path: Sobol initialisation in the unit cube
(bo.py:505-537, utils/core.py:181-193), the initial multi-restart fit (bo.py:611 -> pool.py:268-293), the
WIPV / WIPStd / EI iteration (bo.py:1174-1224, 1226-1390: mc points -> get_next_batch -> evaluate ->
update_gp) and the refit policy of ``update_gp`` (bo.py:620-668, strict ``<`` size classes included).

Run-level resume (bo.py:327-381): ``save=True`` writes ``<save_dir>/<name>_gp.npz`` (the reference's file) plus
``<name>_run.json`` / ``<name>_mc.npz`` (what its results manager keeps: iteration, evaluation count, acquisition
history, convergence state — and, so that a resumed run CONTINUES the interrupted one, the generator state and the
current integration samples); ``BOBE(..., resume=True, resume_file=<save_dir>/<name>)`` picks them up.

Not reproduced: the MPI pool itself (its restart sharding is, over torch.distributed: ``gp_fit``),
NUTS.  The logZ convergence test (bo.py:886-891)
runs on ``bobe_amd.samplers.nested_sampling`` (batched on the GPU GP) instead of dynesty; the loop also stops
on ``max_evals``, ``max_gp_size`` or an acquisition-value threshold.  Integration points come from the
reference's ``'uniform'`` (scrambled Sobol, acquisition.py:476-479) or ``'NS'`` (acquisition.py:473-475) method.
"""
from __future__ import annotations

import json
import math
import os
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


def _factorisable_start(gp: GP, init: np.ndarray) -> np.ndarray:
    """Row 0 of a fit's starts when the surrogate's current hyper-parameters no longer factorise.

    Not in the reference.  Its fit evaluates every start and keeps the best FINITE one (optim.py:325-345); when new points
    have made K at the incumbent numerically singular (``gp.not_pd``: a long-length-scale, large-variance fit of a smooth
    likelihood at the default noise of 1e-8 gets there as N grows) only the uniform random starts are left, and the fit
    jumps to whatever local optimum they find - in the 10-D Rosenbrock run at threshold 0.02 to length scales of 0.03
    and a logZ of +21 (profiles/r04_config5.txt, section 10).  Here the incumbent is first walked back along the kernel
    variance, by factors of four, to the nearest point that does factorise (same length scales, same shape of the
    posterior mean): the fit then starts from the edge of the resolvable region instead of from nowhere."""
    if not getattr(gp, "not_pd", False) or gp.fixed_kernel_variance:
        return init
    k = gp.ndim                                                  # position of log kernel_variance (gp.py:339-355)
    lo = gp.hyperparam_bounds[0][k]
    cands = []
    for step in range(1, 17):
        c = np.array(init)
        c[k] = max(init[k] - step * math.log(4.0), lo)
        cands.append(c)
        if c[k] <= lo:
            break
    for b0 in range(0, len(cands), 8):
        vals = gp.neg_mll_value_and_grad_batch(cands[b0:b0 + 8], want_grad=False)
        for c, (f, _) in zip(cands[b0:b0 + 8], vals):
            if np.isfinite(f):
                log.warning("the surrogate's hyper-parameters no longer factorise at N = %d (kernel variance %.3g): the fit "
                            "starts from kernel variance %.3g instead", gp.npoints, math.exp(init[k]), math.exp(c[k]))
                return c
    return init


def gp_fit(gp: GP, maxiters: int = 1000, n_restarts: int = 8, rng: Optional[np.random.Generator] = None,
           group=None, distributed: bool = True) -> dict:
    """``MPI_Pool.gp_fit`` (pool.py:268-328): x0 row 0 = log(current hp), further rows uniform in the log-bounds;
    fit; adopt the best hyper-parameters (refactors on the GPU).

    With an initialised ``torch.distributed`` group of G > 1 ranks (one process per GPU, every rank running the
    same loop with the same seed) the restarts are split over the ranks the way the reference's MPI pool splits
    them (``np.array_split``, pool.py:298-326): each rank runs its chunk — concurrently, on the evaluation slots of
    its own GPU — then one all-gather of (mll, theta) and max-by-mll; every rank adopts the same theta.
    ``distributed=False`` keeps the fit on this rank (no collective: for calls that not every rank makes)."""
    rng = np.random.default_rng() if rng is None else rng
    n_params = gp.hyperparam_bounds.shape[1]
    init = _factorisable_start(gp, np.log(gp.get_hyperparams()))
    if n_restarts > 1:
        x0 = np.vstack([init, rng.uniform(gp.hyperparam_bounds[0], gp.hyperparam_bounds[1],
                                          size=(n_restarts - 1, n_params))])
    else:
        x0 = np.atleast_2d(init)
    world, rank, coll_dev = dist_info(group, gp.device) if distributed else (1, 0, None)
    if world > 1 and x0.shape[0] > 1:
        lo, hi = shard_bounds(x0.shape[0], world, rank)
        res = gp.fit(x0=x0[lo:hi], maxiter=maxiters) if hi > lo else {"mll": -np.inf, "params": init}
        mll, params = merge_best_fit(res["mll"], res["params"], group=group, device=coll_dev)
        res = {"mll": mll, "params": params}
    else:
        res = gp.fit(x0=x0, maxiter=maxiters)
    gp.update_hyperparams(res["params"])
    return res


def load_gp_file(filename: str, clf: bool, device: int = 0):
    """bo.py:25-43: a ``GP`` or ``GPwithClassifier`` from ``<filename>.npz`` (L and alpha restored without a
    factorisation for the plain GP, gp.py:671-675)."""
    if clf:
        from .clf_gp import GPwithClassifier
        return GPwithClassifier.load(filename, device=device)
    return GP.load(filename, device=device)


def refit_policy(n_train_before: int, n_since_last_fit: int, n_new: int, fit_n_points: int):
    """The size classes of ``update_gp`` (bo.py:632-655) -> (refit, n_restarts, maxiter, points since the last fit).
    Strict '<' on both sides as in the reference: N == 200 and N >= 750 both land in the last branch."""
    n_since = n_since_last_fit + n_new
    if n_train_before < 200:
        refit_threshold, maxiter, n_restarts = min(2, fit_n_points), 1000, 8
    elif 200 < n_train_before < 750:
        refit_threshold, n_restarts, maxiter = fit_n_points, 4, 500
    else:
        refit_threshold, n_restarts, maxiter = max(40, fit_n_points), 4, 200
    return n_since >= refit_threshold, n_restarts, maxiter, n_since


class BOBE:
    """``BOBE(loglikelihood, param_list, param_bounds, ...).run(acq=...)`` with the GP on a MI355X."""

    def __init__(self, loglikelihood: Callable[[np.ndarray], float], param_list: Sequence[str] = None,
                 param_bounds: np.ndarray = None, param_labels=None, likelihood_name: Optional[str] = None,
                 confidence_for_unbounded=0.9999995, gp_kwargs: Optional[dict] = None, n_cobaya_init: int = 4,
                 n_sobol_init: int = 16, init_train_x=None, init_train_y=None, resume: bool = False, resume_file=None,
                 save_dir: str = ".", save: bool = False, save_step: int = 5, optimizer: str = "scipy",
                 acq: str = "WIPV", use_clf: bool = False, clf_type: str = "svm", clf_nsigma_threshold: float = 20,
                 clf_use_size: int = 10, clf_update_step: int = 1, minus_inf: float = -1e10,
                 seed: Optional[int] = None, verbosity: str = "INFO", device: int = 0):
        """Keywords of the reference constructor (bo.py:69-96) plus ``device``.  ``loglikelihood`` must be a callable
        on physical parameters (Cobaya likelihoods belong to the parts that are not built);
        ``save`` writes ``<save_dir>/<likelihood_name>_gp.npz`` (+ the run state) every ``save_step`` iterations
        (bo.py:239); ``resume=True, resume_file=<save_dir>/<likelihood_name>`` continues from those files instead of
        drawing and evaluating an initial design (bo.py:205-206, 327-381; a file that cannot be loaded falls back to a
        fresh start, as there); ``use_clf`` selects ``GPwithClassifier`` (SVM) with the thresholds derived from
        ``clf_nsigma_threshold``."""
        import logging
        if not callable(loglikelihood):
            raise NotImplementedError("only a callable log-likelihood is supported (Cobaya adaptors are out of scope)")
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
        self.save_path = os.path.join(self.save_dir, self.likelihood_name)
        self.fresh_start, self._resume_state = True, None
        if resume and resume_file is not None:
            self._handle_resume(str(resume_file), use_clf)
        if not self.fresh_start:
            return
        # Sobol initial design (bo.py:521-529), optionally after user-supplied points (bo.py:505-519)
        n_sobol = max(2, n_sobol_init)
        sobol = qmc.Sobol(d=self.ndim, scramble=True, seed=self.np_rng).random(n_sobol)
        pts = scale_from_unit(sobol, self.param_bounds)
        vals = self._evaluate(pts)
        if init_train_x is not None and init_train_y is not None:
            pts = np.vstack([np.atleast_2d(np.asarray(init_train_x, dtype=np.float64)), pts])
            vals = np.vstack([np.asarray(init_train_y, dtype=np.float64).reshape(-1, 1), vals])
        kw = dict(optimizer=optimizer)               # everything else: the GP's own defaults, as bo.py:584-605 leaves them
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

    def _handle_resume(self, resume_file: str, use_clf: bool) -> None:
        """bo.py:327-381: the GP from ``<resume_file>_gp.npz`` (tested with one prediction), the run state from
        ``<resume_file>_run.json`` when it is there (a GP file alone resumes at iteration 0 with that training set)."""
        gp_file = resume_file + "_gp"
        try:
            log.info(f"Attempting to resume from file {resume_file}")
            self.gp = load_gp_file(gp_file, use_clf, device=self.device)
            _ = self.gp.predict_mean_single(self.gp.train_x[0])
            log.info(f"Loaded GP with {self.gp.train_x.shape[0]} training points")
        except Exception as e:
            log.error(f"Failed to load GP from file {gp_file}: {e}")
            log.info("Starting a fresh run instead.")
            self.fresh_start = True
            return
        self.fresh_start = False
        try:
            with open(resume_file + "_run.json") as fh:
                st = json.load(fh)
            if int(st.get("gp_training_set_size", -1)) != int(self.gp.npoints):
                raise ValueError(f"run state is for {st.get('gp_training_set_size')} training points, the GP file has "
                                 f"{self.gp.npoints}")
            mc_file = resume_file + "_mc.npz"
            if st.get("has_mc") and os.path.exists(mc_file):
                z = np.load(mc_file, allow_pickle=False)
                st["mc"] = {k: (z[k] if z[k].shape != () else z[k].item()) for k in z.files}
            self._resume_state = st
            log.info(f"Resuming from iteration {st['iteration']} ({st['current_evals']} evaluations so far)")
        except FileNotFoundError:
            log.info("No run state beside the GP file: resuming with its training set at iteration 0")
        except Exception as e:
            log.warning(f"Run state {resume_file}_run.json unusable ({e}): resuming with the GP file alone")

    def _checkpoint(self, state: dict, mc: Optional[dict]) -> None:
        """``<save_path>_gp.npz`` (bo.py:239) and, beside it, the run state a resumed run continues from.  Every file is
        written under a temporary name and moved into place (``os.replace``), the run state LAST: a kill at any moment
        leaves the previous generation readable, and a run state never names a GP file that is newer or half written
        (``_handle_resume`` checks ``gp_training_set_size``)."""
        os.makedirs(self.save_dir, exist_ok=True)
        tmp_gp = self.save_path + "_gp.tmp"                              # (np.savez appends .npz)
        self.gp.save(tmp_gp)
        st = dict(state, rng_state=self.np_rng.bit_generator.state, gp_training_set_size=int(self.gp.npoints),
                  n_points_since_last_fit=int(self.n_points_since_last_fit), timing=dict(self.timing),
                  has_mc=mc is not None)
        tmp_mc = None
        if mc is not None:                                              # (arrays and scalars; acquisition uses 'x' only)
            tmp_mc = self.save_path + "_mc.tmp.npz"
            np.savez(tmp_mc, **{k: np.asarray(v) for k, v in mc.items()
                                if isinstance(v, (np.ndarray, float, int, str, np.generic))})
        tmp = self.save_path + "_run.json.tmp"
        with open(tmp, "w") as fh:
            json.dump(st, fh)
        os.replace(tmp_gp + ".npz", self.save_path + "_gp.npz")
        if tmp_mc is not None:
            os.replace(tmp_mc, self.save_path + "_mc.npz")
        os.replace(tmp, self.save_path + "_run.json")                  # (never a half-written state file)

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
        """bo.py:620-676 — count, decide (``refit_policy``), ``gp.update``, multi-restart fit, classifier retrain."""
        t0 = time.time()
        refit, n_restarts, maxiter, self.n_points_since_last_fit = refit_policy(
            self.gp.train_x.shape[0], self.n_points_since_last_fit, new_pts_u.shape[0], fit_n_points)
        self.gp.update(new_pts_u, new_vals)
        if getattr(self.gp, "not_pd", False):        # the new point made K numerically singular at the old
            refit = True                             # hyper-parameters (NaN factor, like XLA): refit now
        if refit:
            gp_fit(self.gp, n_restarts=n_restarts, maxiters=maxiter, rng=self.np_rng)
            self.n_points_since_last_fit = 0
        self.timing["GP Training"] += time.time() - t0
        if hasattr(self.gp, "train_classifier"):     # bo.py:673-676: the labels move with the best value seen
            t0 = time.time()
            self.gp.train_classifier()
            self.timing["Classifier Training"] = self.timing.get("Classifier Training", 0.0) + time.time() - t0

    def _mc_samples(self, method, num_hmc_warmup, num_hmc_samples, thinning, num_chains, num_mc_samples):
        if method == "NUTS":                         # bo.py:1243-1255: HMC settings of run()
            return get_mc_samples(self.gp, warmup_steps=num_hmc_warmup, num_samples=num_hmc_samples, thinning=thinning,
                                  method="NUTS", num_chains=num_chains, np_rng=self.np_rng)
        return get_mc_samples(self.gp, num_samples=num_mc_samples, method=method, np_rng=self.np_rng)

    def run(self, acq=None, min_evals: int = 200, max_evals: int = 1500, max_gp_size: int = 1200,
            logz_threshold: float = 0.01, convergence_n_iters: int = 1, do_final_ns: bool = False,
            fit_n_points: int = 10, ns_n_points: int = 10, batch_size: int = 1, num_hmc_warmup: int = 512,
            num_hmc_samples: int = 512, mc_points_size: int = 64, thinning: int = 4, num_chains: int = 4,
            mc_points_method: str = "NUTS", zeta_ei: float = 0.01, ei_goal: float = 1e-10,
            num_mc_samples: int = 1024, acq_threshold: Optional[float] = None, verbose: bool = False) -> dict:
        """``BOBE.run`` (bo.py:967-1172): the keywords and defaults of the reference (plus ``num_mc_samples`` for the
        'uniform' / 'NS' integration-point methods and an optional ``acq_threshold`` stop).

        WIPV / WIPStd (bo.py:1226-1385): integration samples once before the loop; per iteration a kriging-believer
        batch, likelihood evaluations, ``update_gp``; when ``ns_n_points`` new evaluations have accumulated past
        ``min_evals`` AND the last acquisition value is <= ``logz_threshold``, nested sampling on the surrogate —
        its equal-weight samples become the next integration samples and (upper - lower)/2 < threshold,
        ``convergence_n_iters`` times in a row, ends the run ("LogZ converged", bo.py:886-934); otherwise fresh
        integration samples.  EI / LogEI (bo.py:1174-1224): one point per iteration, log-EI goal."""
        from .samplers import nested_sampling, resample_equal
        acq = acq if acq is not None else self.default_acq
        if isinstance(acq, (tuple, list)):                       # the reference accepts a tuple of stages: first one
            acq = acq[0]
        acq_fn = _ACQ[acq.lower()]()
        is_wip = acq.lower() in ("wipv", "wipstd")
        acq_hist: List[float] = []
        logz: dict = {}
        ns_samples: Optional[dict] = None
        ns_success = False
        converged, counter, n_since_ns = False, 0, 0
        reason, it = "Max evaluation budget reached", 0          # bo.py:1093
        self.n_points_since_last_fit = 0
        self.timing.setdefault("Nested Sampling", 0.0)
        self.timing.setdefault("MCMC Sampling", 0.0)
        current_evals = self.gp.npoints
        mc_args = (mc_points_method, num_hmc_warmup, num_hmc_samples, thinning, num_chains, num_mc_samples)

        def check_logz(lz) -> bool:                              # bo.py:871-934 (the KL bookkeeping is results-manager work)
            nonlocal counter
            if lz.get("truncated"):                              # the sampler was cut by its call budget: its evidence is a
                log.warning("nested sampling hit its call budget; not testing convergence on a truncated run")
                counter = 0                                      # lower bound, not an estimate (the reference's dynesty run
                return False                                     # would stop the same way, silently)
            delta = (lz["upper"] - lz["lower"]) / 2.0
            if delta < logz_threshold:
                counter += 1
                return counter >= convergence_n_iters
            counter = 0
            return False

        def check_budget() -> Optional[str]:                     # bo.py:757-775
            if current_evals >= max_evals:
                return "Maximum evaluations reached"
            if self.gp.train_x.shape[0] >= max_gp_size:
                return "Maximum GP size reached"
            return None

        mc = None
        rs = self._resume_state
        self._resume_state = None                                # (a second run() on this object starts from its current state)
        if rs is not None and rs.get("acq", acq).lower() == acq.lower():
            # continue the interrupted run (bo.py:337-372: iteration, histories, convergence state)
            it, current_evals, n_since_ns, counter = rs["iteration"], rs["current_evals"], rs["n_since_ns"], rs["counter"]
            acq_hist, logz = list(rs["acq_history"]), dict(rs.get("logz") or {})
            self.n_points_since_last_fit = int(rs.get("n_points_since_last_fit", 0))
            for k_, v_ in (rs.get("timing") or {}).items():
                self.timing[k_] = float(v_)
            self.np_rng.bit_generator.state = rs["rng_state"]
            mc = rs.get("mc")
            if rs.get("converged"):                              # the saved run had already met its stopping rule
                converged, reason = True, rs.get("termination_reason", "LogZ converged")
        if is_wip and mc is None:
            t0 = time.time()
            mc = self._mc_samples(*mc_args)
            self.timing["MCMC Sampling"] += time.time() - t0

        def run_state():
            return {"acq": acq, "iteration": it, "current_evals": int(current_evals), "n_since_ns": int(n_since_ns),
                    "counter": int(counter), "acq_history": [float(a) for a in acq_hist], "converged": bool(converged),
                    "termination_reason": reason,
                    "logz": {k_: (float(v_) if isinstance(v_, (int, float, np.floating, np.integer)) else v_)
                             for k_, v_ in logz.items() if isinstance(v_, (int, float, bool, np.floating, np.integer))}}
        while not converged:
            it += 1
            t0 = time.time()
            if is_wip:
                n_since_ns += batch_size
                ns_flag = n_since_ns >= ns_n_points and current_evals >= min_evals
                kwargs = {"mc_samples": mc, "mc_points_size": mc_points_size}
                new_u, vals = acq_fn.get_next_batch(self.gp, n_batch=batch_size, acq_kwargs=kwargs, n_restarts=1,
                                                    maxiter=100, early_stop_patience=10, rng=self.np_rng)  # bo.py:1274
                n_new = batch_size
            else:
                kwargs = {"zeta": zeta_ei, "best_y": float(np.max(self.gp.train_y)) if self.gp.train_y.size else 0.0}
                new_u, vals = acq_fn.get_next_batch(self.gp, n_batch=1, acq_kwargs=kwargs, n_restarts=50,
                                                    maxiter=1000, early_stop_patience=50, rng=self.np_rng)  # bo.py:1194
                n_new = 1
            new_u = np.atleast_2d(new_u)
            vals = np.atleast_1d(vals)
            self.timing["Acquisition Optimization"] += time.time() - t0
            acq_hist.append(float(np.mean(vals)))
            new_vals = self._evaluate(scale_from_unit(new_u, self.param_bounds))
            current_evals += n_new                               # a proposal the GP rejects as a duplicate still counts
            self.update_gp(new_u, new_vals, fit_n_points)
            if verbose:
                log.info(f"Iteration {it}: N={self.gp.npoints} acq={acq_hist[-1]:.3e}")
            if is_wip:
                if ns_flag and float(vals[-1]) <= logz_threshold:            # bo.py:1283-1311
                    t0 = time.time()
                    ns_samples, lz, ns_success = nested_sampling(self.gp, mode="convergence", dlogz=0.01,
                                                                 equal_weights=False, rng=self.np_rng)
                    self.timing["Nested Sampling"] += time.time() - t0
                    if ns_success:
                        eq_x, eq_l = resample_equal(ns_samples["x"], ns_samples["logl"], ns_samples["weights"],
                                                    rng=self.np_rng)
                        mc = {"x": eq_x, "logl": eq_l, "weights": np.ones(eq_x.shape[0]), "method": "NS",
                              "best": ns_samples["best"]}
                        logz = lz
                        converged = check_logz(lz)
                        if converged:
                            reason = "LogZ converged"
                    n_since_ns = 0
                else:                                                        # bo.py:1313-1324
                    t0 = time.time()
                    mc = self._mc_samples(*mc_args)
                    self.timing["MCMC Sampling"] += time.time() - t0
                if acq_threshold is not None and acq_hist[-1] <= acq_threshold and not converged:
                    reason = "Acquisition threshold reached"
                    break
            else:                                                            # bo.py:838-866, 1208-1218
                goal_val = float(vals[-1])
                if acq.lower() == "ei":
                    goal_val = float(np.log(goal_val + 1e-100))
                if goal_val < np.log(ei_goal):
                    counter += 1
                    if counter >= convergence_n_iters:
                        converged, reason = True, f"{acq_fn.name.upper()} goal reached"
                else:
                    counter = 0
            if self.save and it % self.save_step == 0:
                self._checkpoint(run_state(), mc if is_wip else None)
            if converged:
                break
            budget = check_budget()
            if budget is not None:
                reason = budget
                break
        if self.save:
            # the state a resumed run continues from is the state at the END OF THE LOOP: what follows (a final fit and
            # nested sampling, the result samples) draws from the generator but is not part of the iteration
            self._checkpoint(run_state(), mc if is_wip else None)
        if is_wip and do_final_ns and not converged:                         # bo.py:1345-1366
            t0 = time.time()
            gp_fit(self.gp, n_restarts=4, maxiters=500, rng=self.np_rng)
            self.timing["GP Training"] += time.time() - t0
            t0 = time.time()
            ns_samples, lz, ns_success = nested_sampling(self.gp, mode="convergence", dlogz=0.01, rng=self.np_rng)
            self.timing["Nested Sampling"] += time.time() - t0
            if ns_success:
                logz = lz
                if check_logz(lz):
                    converged, reason = True, "LogZ converged"
        samples: dict = {}
        if is_wip:                                                           # bo.py:1368-1385
            if ns_samples is not None and ns_success:
                x_u, weights, logl = ns_samples["x"], ns_samples["weights"], ns_samples["logl"]
            else:
                t0 = time.time()
                hm = get_mc_samples(self.gp, warmup_steps=512, num_samples=2000 * self.ndim, thinning=4, method="NUTS",
                                    np_rng=self.np_rng)
                self.timing["MCMC Sampling"] += time.time() - t0
                x_u = hm["x"]
                weights = hm["weights"] if "weights" in hm else np.ones(hm["x"].shape[0])
                logl = hm["logp"] if "logp" in hm else hm.get("logl")
            samples = {"x": scale_from_unit(np.asarray(x_u), self.param_bounds), "weights": np.asarray(weights),
                       "logl": np.asarray(logl)}
        if self.save and is_wip and do_final_ns:                             # (the final fit changed the hyper-parameters)
            self.gp.save(self.save_path + "_gp.tmp")
            os.replace(self.save_path + "_gp.tmp.npz", self.save_path + "_gp.npz")
        y = self.gp.train_y * self.gp.y_std + self.gp.y_mean
        ibest = int(np.argmax(y))
        best_x = scale_from_unit(self.gp.train_x[ibest], self.param_bounds)
        manager = {"likelihood_name": self.likelihood_name, "param_list": self.param_list,
                   "param_labels": self.param_labels, "acquisition_history": list(acq_hist),
                   "timing": dict(self.timing), "converged": converged, "termination_reason": reason,
                   "gp_training_set_size": int(self.gp.npoints)}
        # keys of the reference's results dict (bo.py:827-836); the rest are conveniences of this driver
        return {"gp": self.gp, "likelihood": self.loglikelihood, "results_manager": manager,
                "best_val": float(y[ibest, 0]), "best_pt": best_x, "logz": logz, "termination_reason": reason,
                "samples": samples,
                "best_x": best_x, "n_evals": int(self.gp.npoints), "acq_history": acq_hist, "timing": dict(self.timing),
                "lengthscales": np.array(self.gp.lengthscales), "kernel_variance": float(self.gp.kernel_variance),
                "converged": converged}
