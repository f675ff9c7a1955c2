"""BO-loop counterpart (SURVEY.md 8f "next" row 1) — a compact driver around the GPU GP.

Mirrors the parts of ``BOBE/bo.py`` that *call* the hot path: Sobol initialisation in the unit cube
(bo.py:505-537, utils/core.py:181-193), the initial multi-restart fit (bo.py:611 -> pool.py:268-293), the
WIPV / WIPStd / EI iteration (bo.py:1174-1224, 1226-1390: mc points -> get_next_batch -> evaluate ->
update_gp) and the refit policy of ``update_gp`` (bo.py:620-668, strict ``<`` size classes included).

Run-level resume (bo.py:327-381): ``save=True`` (the default, as there) writes ``<save_dir>/<name>_gp.npz`` (the
reference's file) plus ``<name>_run.json`` naming one complete generation ``<name>_gp.<g>.npz`` / ``<name>_mc.<g>.npz``
(what its results manager keeps: iteration, evaluation count, acquisition history, convergence state — and, so that a
resumed run CONTINUES the interrupted one, the generator state and the current integration samples);
``BOBE(..., resume=True, resume_file=<save_dir>/<name>)`` picks them up.

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
from .likelihood import Likelihood
from .utils import (get_logger, get_numpy_rng, scale_from_unit, scale_to_unit, set_global_seed, setup_logging,
                    update_verbosity)

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
           use_pool: bool = True, *, group=None, distributed: bool = True, factorisable_start: bool = True) -> dict:
    """``MPI_Pool.gp_fit`` (pool.py:268-328): x0 row 0 = log(current hp), further rows uniform in the log-bounds;
    fit; adopt the best hyper-parameters (refactors on the GPU).

    With an initialised ``torch.distributed`` group of G > 1 ranks (one process per GPU, every rank running the
    same loop with the same seed) the restarts are split over the ranks the way the reference's MPI pool splits
    them (``np.array_split``, pool.py:298-326): each rank runs its chunk — concurrently, on the evaluation slots of
    its own GPU — then one all-gather of (mll, theta) and max-by-mll; every rank adopts the same theta.
    ``use_pool=False`` (the reference's keyword, pool.py:239) or ``distributed=False`` keeps the fit on this rank (no
    collective: for calls that not every rank makes).  ``factorisable_start=False`` keeps row 0 = log(current hp) whatever
    happens at it, as the reference does (pool.py:277-284; ``_factorisable_start`` above)."""
    distributed = bool(distributed and use_pool)
    rng = np.random.default_rng() if rng is None else rng
    n_params = gp.hyperparam_bounds.shape[1]
    init = np.log(gp.get_hyperparams())
    if factorisable_start:
        init = _factorisable_start(gp, init)
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


def load_gp_statedict(state_dict: dict, clf: bool, device: int = 0):
    """bo.py:45-65: the same from a state dictionary (what the reference's MPI workers are sent, pool.py:109-117)."""
    if clf:
        from .clf_gp import GPwithClassifier
        return GPwithClassifier.from_state_dict(state_dict, device=device)
    return GP.from_state_dict(state_dict, device=device)


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
    """``BOBE(loglikelihood, param_list, param_bounds, ...).run(...)`` with the GP on a MI355X: the constructor, ``run`` and
    the helper methods a script can reach (``update_gp``, ``get_next_batch``, ``evaluate_likelihood``,
    ``check_max_evals_and_gpsize``, ``check_convergence_ei`` / ``_logz``, ``finalise_results``, ``run_EI`` / ``run_WIPStd`` /
    ``run_WIPV`` / ``run_weighted_integrated_posterior``) carry the reference's names, argument order and defaults
    (bo.py:69-96, 621, 681, 707, 758, 779, 838, 869, 967-984, 1174, 1226, 1392, 1396; pinned by
    tests/golden/reference_signatures.json); what this driver adds is keyword-only and comes after them."""

    def __init__(self, loglikelihood: Callable[[np.ndarray], float], param_list: Sequence[str] = None,
                 param_bounds: np.ndarray = None, param_labels=None, likelihood_name: Optional[str] = None,
                 confidence_for_unbounded=0.9999995, gp_kwargs: Optional[dict] = None, n_cobaya_init: int = 4,
                 n_sobol_init: int = 16, init_train_x=None, init_train_y=None, resume: bool = False, resume_file=None,
                 save_dir: str = ".", save: bool = True, save_step: int = 5, optimizer: str = "scipy",
                 acq: str = "WIPV", use_clf: bool = False, clf_type: str = "svm", clf_nsigma_threshold: float = 20,
                 clf_use_size: int = 10, clf_update_step: int = 1, minus_inf: float = -1e10,
                 seed: Optional[int] = None, verbosity: str = "INFO", *, device: int = 0,
                 factorisable_start: bool = True):
        """Keywords, order and defaults of the reference constructor (bo.py:69-96) plus the keyword-only ``device`` and
        ``factorisable_start`` (deviation (viii), DESIGN.md 8: a fit whose incumbent no longer factorises starts from the
        nearest kernel variance that does; False = the reference's starts).  The surrogate is built with the GP's own
        defaults except one: ``pivot_floor_ulp = 64`` (deviation (vii), the rank test; the ``GP`` class by itself keeps the
        reference's sign-only rule) - ``gp_kwargs={'pivot_floor_ulp': 0}`` switches it off for the run.
        ``loglikelihood`` must be a callable on physical parameters (Cobaya likelihoods belong to the parts that are not
        built).  ``acq`` is recorded among the settings only, as in the reference (bo.py:314): what runs is
        ``run``'s own ``acq``.  ``save`` (default True, bo.py:84) writes ``<save_dir>/<likelihood_name>_gp.npz`` after the
        initial fit (bo.py:239) and, with the run state beside it, every ``save_step`` iterations;
        ``resume=True, resume_file=<save_dir>/<likelihood_name>`` continues from those files instead of drawing and
        evaluating an initial design (bo.py:205-206, 327-381; a file that cannot be loaded falls back to a fresh start, as
        there); ``use_clf`` selects ``GPwithClassifier`` (SVM) with the thresholds derived from ``clf_nsigma_threshold``."""
        import logging
        if not logging.getLogger("bobe_amd").handlers and not logging.getLogger().handlers:
            setup_logging(verbosity)              # (the reference installs its handlers when the package is imported,
        else:                                     # BOBE/__init__.py:37-39; here: on first use, unless the host program has
            update_verbosity(verbosity)           # configured logging itself)                                    bo.py:179
        if str(optimizer).lower() not in ("optax", "scipy"):                 # bo.py:299-300
            raise ValueError("optimizer must be either 'optax' or 'scipy'")
        self.loglikelihood = self._prepare_likelihood(loglikelihood, param_list, param_bounds, param_labels,
                                                      likelihood_name, confidence_for_unbounded, minus_inf)
        self.param_list = list(self.loglikelihood.param_list)
        self.param_labels = list(self.loglikelihood.param_labels)
        self.likelihood_name = self.loglikelihood.name
        self.output_file = self.likelihood_name                             # bo.py:290
        self.param_bounds = np.asarray(self.loglikelihood.param_bounds, dtype=np.float64)      # (2, ndim)
        self.ndim = len(self.param_list)
        set_global_seed(seed)                                               # bo.py:286-287: ONE generator for the whole run,
        self.np_rng = get_numpy_rng()                                       # the package's global one
        self.minus_inf = float(minus_inf)
        self.optimizer = optimizer
        self.device = device
        # No MPI pool: one process per GPU, every rank runs this loop with the same seed.  Rank 0 of the
        # initialised torch.distributed group is the main process (bo.py:97-101 asks the pool the same) and the ONLY one that
        # writes files: the checkpoint generations, the run state and the reference's <name>_gp.npz.
        self.is_main, self.is_mpi = dist_info(None, device)[1] == 0, False
        self.save, self.save_dir, self.save_step = bool(save) and self.is_main, save_dir, max(1, int(save_step))
        self.factorisable_start = bool(factorisable_start)
        self.pivot_floor_ulp = float((gp_kwargs or {}).get("pivot_floor_ulp", 64.0))
        self.settings = {"n_cobaya_init": n_cobaya_init, "n_sobol_init": n_sobol_init, "acq": acq, "use_clf": use_clf,
                         "clf_type": clf_type, "clf_nsigma_threshold": clf_nsigma_threshold, "minus_inf": minus_inf,
                         "seed": seed}                                      # bo.py:311-320
        self.timing: Dict[str, float] = {"GP Training": 0.0, "Acquisition Optimization": 0.0,
                                         "True Objective Evaluations": 0.0}
        self.n_points_since_last_fit = 0
        self.n_points_since_last_ns = 0
        # run()'s settings at their defaults, so that the helper methods can be called before (or without) run()
        self.fit_n_points, self.ns_n_points, self.batch_size = 10, 10, 4      # (run() stores its own, bo.py:1082-1084)
        self.min_evals, self.max_evals, self.max_gp_size, self.logz_threshold = 200, 1500, 1200, 0.01
        self.convergence_n_iters, self.ei_goal_log, self.do_final_ns = 1, float(np.log(1e-10)), False
        self.num_hmc_warmup, self.num_hmc_samples, self.mc_points_size = 512, 512, 64
        self.hmc_thinning, self.hmc_num_chains, self.mc_points_method, self.zeta_ei = 4, 4, "NUTS", 0.01
        self.num_mc_samples, self.acq_threshold, self.verbose = 1024, None, False
        self.min_delta_seen = np.inf
        self.current_iteration = 0
        self.start_iteration = 0
        self.acquisition = _ACQ.get(str(acq).lower(), WIPV)(optimizer=optimizer)   # (run() installs its own, bo.py:1147)
        self.acquisition_history: List[float] = []
        self.gp_hyperparam_history: List[dict] = []
        self.kl_history: List[dict] = []
        self.convergence_history: List[dict] = []
        self.prev_samples = None                                            # bo.py:243
        self.results_dict: dict = {}
        self.samples_dict: dict = {}
        self.mc_samples: Optional[dict] = None
        self.ns_samples: Optional[dict] = None
        self._ns_success = False
        self.converged, self.convergence_counter = False, 0
        self.termination_reason = "Max evaluation budget reached"
        self.save_path = os.path.join(self.save_dir, self.likelihood_name)
        self._ckpt_gen = 0
        self._current_evals = 0
        self.fresh_start, self._resume_state = True, None
        if resume and resume_file is not None:
            self._handle_resume(str(resume_file), use_clf)
        if self.fresh_start:
            self._handle_fresh_start(n_sobol_init, init_train_x, init_train_y, use_clf, clf_type, clf_use_size,
                                     clf_update_step, clf_nsigma_threshold, optimizer, gp_kwargs)
        # best point so far (bo.py:217-236)
        y = self.gp.train_y * self.gp.y_std + self.gp.y_mean
        ibest = int(np.argmax(y))
        self.best_f = float(y.reshape(-1)[ibest])
        self.best_pt = scale_from_unit(self.gp.train_x[ibest], self.param_bounds).reshape(-1)
        self.best = {name: f"{float(val):.6f}" for name, val in zip(self.param_list, self.best_pt)}
        self.best_pt_iteration = self.start_iteration
        if self.save and self.fresh_start:                                  # bo.py:239
            os.makedirs(self.save_dir, exist_ok=True)
            self._save_gp_file()

    def _prepare_likelihood(self, loglikelihood, param_list, param_bounds, param_labels, likelihood_name,
                            confidence_for_unbounded, minus_inf) -> Likelihood:
        """bo.py:249-280: a ``Likelihood`` as it is, a callable wrapped into one.  A Cobaya YAML path / info dict would
        need the Cobaya adaptor, which is not built."""
        if isinstance(loglikelihood, Likelihood):
            return loglikelihood
        if isinstance(loglikelihood, (str, dict)):
            raise NotImplementedError("Cobaya likelihoods (a YAML path or an info dict) need the Cobaya adaptor, which is "
                                      "outside this build's scope; pass a callable or a Likelihood")
        if callable(loglikelihood):
            if param_list is None:
                raise ValueError("param_list is required with a callable log-likelihood")
            return Likelihood(loglikelihood=loglikelihood, param_list=list(param_list), param_bounds=param_bounds,
                              param_labels=param_labels, name=likelihood_name, minus_inf=minus_inf)
        raise ValueError("loglikelihood must be one of: callable, string (Cobaya YAML path), dict (Cobaya info), or "
                         "Likelihood instance")

    def _handle_fresh_start(self, n_sobol_init, init_train_x, init_train_y, use_clf, clf_type, clf_use_size,
                            clf_update_step, clf_nsigma_threshold, optimizer, gp_kwargs) -> None:
        """bo.py:383-414: Sobol initial design (bo.py:521-529), optionally after user-supplied points (bo.py:505-519),
        the surrogate (bo.py:571-612) and its first fit (bo.py:611)."""
        n_sobol = max(2, n_sobol_init)
        sobol = qmc.Sobol(d=self.ndim, scramble=True, seed=self.np_rng).random(n_sobol)
        pts = scale_from_unit(sobol, self.param_bounds)
        vals = self._evaluate(pts)
        if init_train_x is not None and init_train_y is not None:
            pts = np.vstack([np.atleast_2d(np.asarray(init_train_x, dtype=np.float64)), pts])
            vals = np.vstack([np.asarray(init_train_y, dtype=np.float64).reshape(-1, 1), vals])
        # everything else: the GP's own defaults, as bo.py:584-605 leaves them
        kw = dict(optimizer=optimizer, pivot_floor_ulp=self.pivot_floor_ulp)
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
                                       device=self.device, **kw)
        else:
            self.gp = GP(x_u, vals, param_names=self.param_list, device=self.device, **kw)
        gp_fit(self.gp, n_restarts=4, maxiters=500, rng=self.np_rng,        # bo.py:611
               factorisable_start=self.factorisable_start)
        self.timing["GP Training"] += time.time() - t0

    # ------------------------------------------------------------------ files
    def _handle_resume(self, resume_file: str, use_clf: bool) -> None:
        """bo.py:327-381.  ``<resume_file>_run.json``, when it is there, names the generation of files it belongs to
        (``_checkpoint``); the GP comes from that generation's file, tested with one prediction.  Without a usable run
        state the reference's own file ``<resume_file>_gp.npz`` resumes at iteration 0 with that training set."""
        st = None
        try:
            with open(resume_file + "_run.json") as fh:
                st = json.load(fh)
        except FileNotFoundError:
            pass
        except Exception as e:
            log.warning(f"Run state {resume_file}_run.json unusable ({e}): resuming with the GP file alone")
        base = os.path.dirname(resume_file)
        if st is not None and st.get("gp_file"):
            try:
                log.info(f"Attempting to resume from file {resume_file} (generation {st.get('generation')})")
                gp = load_gp_file(os.path.join(base, st["gp_file"])[:-len(".npz")], use_clf, device=self.device)
                _ = gp.predict_mean_single(gp.train_x[0])
                if int(st.get("gp_training_set_size", -1)) != int(gp.npoints):
                    raise ValueError(f"run state is for {st.get('gp_training_set_size')} training points, its GP file has "
                                     f"{gp.npoints}")
                if st.get("mc_file"):
                    z = np.load(os.path.join(base, st["mc_file"]), allow_pickle=False)
                    st["mc"] = {k: (z[k] if z[k].shape != () else z[k].item()) for k in z.files}
                gp.pivot_floor_ulp = self.pivot_floor_ulp               # (a run's setting, not part of the reference's file)
                self.gp, self.fresh_start, self._resume_state = gp, False, st
                self._ckpt_gen = int(st.get("generation", 0))
                self.start_iteration = int(st.get("iteration", 0))
                log.info(f"Loaded GP with {gp.train_x.shape[0]} training points; resuming from iteration "
                         f"{st['iteration']} ({st['current_evals']} evaluations so far)")
                return
            except Exception as e:
                log.warning(f"Run state {resume_file}_run.json does not lead to a usable generation ({e}): resuming with "
                            "the GP file alone")
        gp_file = resume_file + "_gp"
        try:
            log.info(f"Attempting to resume from file {resume_file}")
            self.gp = load_gp_file(gp_file, use_clf, device=self.device)
            _ = self.gp.predict_mean_single(self.gp.train_x[0])
            self.gp.pivot_floor_ulp = self.pivot_floor_ulp
            log.info(f"Loaded GP with {self.gp.train_x.shape[0]} training points")
        except Exception as e:
            log.error(f"Failed to load GP from file {gp_file}: {e}")
            log.info("Starting a fresh run instead.")
            self.fresh_start = True
            return
        self.fresh_start = False
        log.info("No usable run state beside the GP file: resuming with its training set at iteration 0")

    def _save_gp_file(self, source: Optional[str] = None) -> None:
        """The reference's file ``<save_path>_gp.npz`` (bo.py:239; loadable by its ``load_gp_file``), moved into place
        under its final name only when complete."""
        tmp = f"{self.save_path}_gp.tmp{os.getpid()}"                     # (np.savez appends .npz)
        if source is None:
            self.gp.save(tmp)
        else:
            import shutil
            shutil.copyfile(source, tmp + ".npz")
        os.replace(tmp + ".npz", self.save_path + "_gp.npz")

    def _checkpoint(self, state: dict, mc: Optional[dict]) -> None:
        """One generation of run files: ``<save_path>_gp.<g>.npz`` and ``<save_path>_mc.<g>.npz`` first, then
        ``<save_path>_run.json`` - replaced atomically, LAST, and naming the two files it belongs with.  A kill at any
        moment therefore leaves a run state whose generation is complete on disk: the previous one until the replace,
        the new one after it (older generations are pruned only then).  ``<save_path>_gp.npz``, the file the reference
        writes and reads (bo.py:239, 25-43), is refreshed after that from the generation's file."""
        os.makedirs(self.save_dir, exist_ok=True)
        self._ckpt_gen += 1
        g = self._ckpt_gen
        name = os.path.basename(self.save_path)
        gp_file = f"{name}_gp.{g}.npz"
        self.gp.save(os.path.join(self.save_dir, gp_file)[:-len(".npz")])
        mc_file = None
        if mc is not None:                                              # (arrays and scalars; acquisition uses 'x' only)
            mc_file = f"{name}_mc.{g}.npz"
            np.savez(os.path.join(self.save_dir, mc_file),
                     **{k: np.asarray(v) for k, v in mc.items()
                        if isinstance(v, (np.ndarray, float, int, str, np.generic))})
        st = dict(state, rng_state=self.np_rng.bit_generator.state, gp_training_set_size=int(self.gp.npoints),
                  n_points_since_last_fit=int(self.n_points_since_last_fit), timing=dict(self.timing),
                  generation=g, gp_file=gp_file, mc_file=mc_file)
        tmp = f"{self.save_path}_run.json.tmp{os.getpid()}"
        with open(tmp, "w") as fh:
            json.dump(st, fh)
        os.replace(tmp, self.save_path + "_run.json")                  # (never a half-written state file)
        self._save_gp_file(os.path.join(self.save_dir, gp_file))
        for f in os.listdir(self.save_dir):                             # prune the generations before this one
            for stem in (name + "_gp.", name + "_mc."):
                if f.startswith(stem) and f.endswith(".npz") and f[len(stem):-len(".npz")].isdigit() \
                        and int(f[len(stem):-len(".npz")]) < g:
                    try:
                        os.remove(os.path.join(self.save_dir, f))
                    except OSError:
                        pass

    def _run_state(self) -> dict:
        """What a resumed run continues from (bo.py:337-372: iteration, histories, convergence state)."""
        num = (int, float, np.floating, np.integer)
        return {"acq": self.acquisition.name if self.acquisition is not None else None, "stage": int(getattr(self, "_stage", 0)),
                "iteration": int(self.current_iteration), "current_evals": int(self._current_evals),
                "n_since_ns": int(self.n_points_since_last_ns), "counter": int(self.convergence_counter),
                "acq_history": [float(a) for a in self.acquisition_history], "converged": bool(self.converged),
                "termination_reason": self.termination_reason,
                "logz": {k: (v if isinstance(v, bool) else float(v)) for k, v in (self.results_dict.get("logz") or {}).items()
                         if isinstance(v, num + (bool,))}}

    # ------------------------------------------------------------------ run helper methods (bo.py:617-934)
    def _evaluate(self, pts: np.ndarray) -> np.ndarray:
        """The likelihood at every row of ``pts`` (``pool.run_map_objective`` in serial mode, pool.py:330-352): the
        ``Likelihood``'s safe evaluation (likelihood.py:61-83), NaN / exceptions / -inf -> minus_inf."""
        t0 = time.time()
        out = np.array([self.loglikelihood(np.array(p)) for p in pts], dtype=np.float64).reshape(-1, 1)
        self.timing["True Objective Evaluations"] += time.time() - t0
        return out

    def update_gp(self, new_pts_u, new_vals, step=0, verbose=True) -> None:
        """bo.py:621-676 — count, decide (``refit_policy`` with the ``fit_n_points`` that ``run`` stored, bo.py:1083),
        ``gp.update``, multi-restart fit, hyper-parameter tracking, classifier retrain."""
        t0 = time.time()
        new_pts_u = np.atleast_2d(np.asarray(new_pts_u, dtype=np.float64))
        refit, n_restarts, maxiter, self.n_points_since_last_fit = refit_policy(
            self.gp.train_x.shape[0], self.n_points_since_last_fit, new_pts_u.shape[0], self.fit_n_points)
        self.gp.update(new_pts_u, new_vals)
        if getattr(self.gp, "not_pd", False):        # the new point made K numerically singular at the old
            refit = True                             # hyper-parameters (NaN factor, like XLA): refit now
        if refit:
            if verbose:
                log.info(f"Refitting GP hyperparameters with {self.gp.train_x.shape[0]} training points ")
            gp_fit(self.gp, n_restarts=n_restarts, maxiters=maxiter, rng=self.np_rng,
                   factorisable_start=self.factorisable_start)
            self.n_points_since_last_fit = 0
        self.timing["GP Training"] += time.time() - t0
        self.gp_hyperparam_history.append({"iteration": int(step), "lengthscales": [float(v) for v in self.gp.lengthscales],
                                           "kernel_variance": float(self.gp.kernel_variance)})
        if hasattr(self.gp, "train_classifier"):     # bo.py:673-676: the labels move with the best value seen
            t0 = time.time()
            self.gp.train_classifier()
            self.timing["Classifier Training"] = self.timing.get("Classifier Training", 0.0) + time.time() - t0

    def get_next_batch(self, acq_kwargs, n_batch, n_restarts, maxiter, early_stop_patience, step, verbose=True):
        """bo.py:681-705: the acquisition's kriging-believer batch + the mean acquisition value in the history."""
        t0 = time.time()
        new_pts_u, acq_vals = self.acquisition.get_next_batch(gp=self.gp, n_batch=n_batch, acq_kwargs=acq_kwargs,
                                                              n_restarts=n_restarts, maxiter=maxiter,
                                                              early_stop_patience=early_stop_patience, rng=self.np_rng)
        self.timing["Acquisition Optimization"] += time.time() - t0
        acq_val = float(np.mean(acq_vals))
        if verbose:
            log.debug(f"Mean acquisition value {acq_val:.4e} at new points")
        self.acquisition_history.append(acq_val)
        return new_pts_u, acq_vals

    def evaluate_likelihood(self, new_pts_u, step, verbose=True):
        """bo.py:707-756: likelihood values (n, 1) at unit-cube points; keeps the best point seen."""
        new_pts_u = np.atleast_2d(np.asarray(new_pts_u, dtype=np.float64))
        new_pts = scale_from_unit(new_pts_u, self.param_bounds)
        new_vals = self._evaluate(new_pts)
        ibest = int(np.argmax(new_vals))
        if float(new_vals[ibest, 0]) > self.best_f:
            self.best_f = float(new_vals[ibest, 0])
            self.best_pt = new_pts[ibest].reshape(-1)
            self.best = {name: f"{float(val):.6f}" for name, val in zip(self.param_list, self.best_pt)}
            self.best_pt_iteration = step
        return new_vals

    def check_max_evals_and_gpsize(self, current_evals) -> bool:
        """bo.py:758-777."""
        if current_evals >= self.max_evals:
            self.termination_reason = "Maximum evaluations reached"
        elif self.gp.train_x.shape[0] >= self.max_gp_size:
            self.termination_reason = "Maximum GP size reached"
        else:
            return False
        self.results_dict["termination_reason"] = self.termination_reason
        return True

    def finalise_results(self) -> None:
        """bo.py:779-836: the results dictionary - the reference's eight keys first (its results manager is a plain dict
        here: histories, timing, settings), then this driver's conveniences."""
        clf = hasattr(self.gp, "train_classifier")
        gp_info = {"gp_training_set_size": int(self.gp.train_x.shape[0]), "gp_final_best_loglike": float(self.best_f),
                   "classifier_used": bool(getattr(self.gp, "use_clf", False)) if clf else False,
                   "classifier_type": str(self.gp.clf_type) if clf else None,
                   "classifier_training_set_size": int(getattr(self.gp, "clf_data_size", 0)) if clf else 0}
        logz = self.results_dict.get("logz", {})
        if not logz:
            log.warning("No logz information found, nested sampling has not been run yet.")
        manager = {"likelihood_name": self.likelihood_name, "param_list": self.param_list,
                   "param_labels": self.param_labels, "settings": dict(self.settings),
                   "acquisition_history": list(self.acquisition_history),
                   "gp_hyperparams": list(self.gp_hyperparam_history), "kl_divergences": list(self.kl_history),
                   "convergence_history": list(self.convergence_history), "timing": dict(self.timing),
                   "converged": bool(self.converged), "termination_reason": self.termination_reason, "gp_info": gp_info,
                   "gp_training_set_size": int(self.gp.npoints)}
        self.results_dict = {"gp": self.gp, "likelihood": self.loglikelihood, "results_manager": manager,
                             "best_val": float(self.best_f), "best_pt": self.best_pt, "logz": logz,
                             "termination_reason": self.termination_reason, "samples": self.samples_dict or {},
                             # conveniences of this driver
                             "best_x": self.best_pt, "n_evals": int(self.gp.npoints),
                             "acq_history": list(self.acquisition_history), "timing": dict(self.timing),
                             "lengthscales": np.array(self.gp.lengthscales),
                             "kernel_variance": float(self.gp.kernel_variance), "converged": bool(self.converged)}

    def check_convergence_ei(self, step, acq_val) -> bool:
        """bo.py:838-867: log EI below log(ei_goal), ``convergence_n_iters`` times in a row."""
        acq_val = float(np.asarray(acq_val).reshape(-1)[-1])
        if self.acquisition.name.lower() == "ei":
            acq_val = float(np.log(acq_val + 1e-100))
        if acq_val < self.ei_goal_log:
            self.convergence_counter += 1
            return self.convergence_counter >= self.convergence_n_iters
        self.convergence_counter = 0
        return False

    def check_convergence_logz(self, step, logz_dict, equal_samples, equal_logl, verbose=True, save_checkpoint=True) -> bool:
        """bo.py:869-961: (upper - lower)/2 < ``logz_threshold``, ``convergence_n_iters`` times in a row; the Gaussian KL
        divergence between successive posterior samples is recorded beside it; a new smallest half-width saves
        ``<save_path>_checkpoint_gp.npz`` when files are written at all.  Not in the reference: a nested-sampling run cut
        by its call budget is never taken as evidence of convergence."""
        if logz_dict.get("truncated"):                           # the sampler was cut by its call budget: its evidence is a
            log.warning("nested sampling hit its call budget; not testing convergence on a truncated run")
            self.convergence_counter = 0                         # lower bound, not an estimate (the reference's dynesty run
            return False                                         # would stop the same way, silently)
        delta = (logz_dict["upper"] - logz_dict["lower"]) / 2.0
        delta_crosscheck = float(logz_dict.get("std", 0.0))
        below = delta < self.logz_threshold
        eq = scale_from_unit(np.atleast_2d(np.asarray(equal_samples, dtype=np.float64)), self.param_bounds)
        if self.prev_samples is not None and eq.shape[0] > self.ndim and self.prev_samples["x"].shape[0] > self.ndim:
            from .utils import kl_divergence_gaussian
            a = self.prev_samples["x"]
            kl = kl_divergence_gaussian(np.mean(a, axis=0), np.atleast_2d(np.cov(a, rowvar=False)),
                                        np.mean(eq, axis=0), np.atleast_2d(np.cov(eq, rowvar=False)))
            self.kl_history.append(dict(kl, iteration=int(step)))
            if verbose:
                log.info(f"Successive KL: symmetric={kl.get('symmetric', 0):.4f}")
        self.prev_samples = {"x": eq, "logl": np.asarray(equal_logl)}
        self.convergence_history.append({"iteration": int(step), "delta": float(delta), "converged": bool(below),
                                         "threshold": float(self.logz_threshold)})
        if verbose:
            log.info(f"Convergence check: delta = {delta:.4f}, step = {step}, threshold = {self.logz_threshold}")
        if below:
            self.convergence_counter += 1
            converged = self.convergence_counter >= self.convergence_n_iters
        else:
            self.convergence_counter = 0
            converged = False
        if delta < self.min_delta_seen and delta_crosscheck < 1.0 and save_checkpoint:
            self.min_delta_seen = delta
            if not converged and self.save:
                os.makedirs(self.save_dir, exist_ok=True)
                self.gp.save(self.save_path + "_checkpoint_gp")
        return converged

    # ------------------------------------------------------------------ main run methods (bo.py:963-1397)
    def run(self, acq="wipstd", min_evals: int = 200, max_evals: int = 1500, max_gp_size: int = 1200,
            logz_threshold: float = 0.01, convergence_n_iters: int = 1, ei_goal: float = 1e-10,
            do_final_ns: bool = False, fit_n_points: int = 10, batch_size: int = 4, ns_n_points: int = 10,
            num_hmc_warmup: int = 512, num_hmc_samples: int = 512, mc_points_size: int = 64, thinning: int = 4,
            num_chains: int = 4, mc_points_method: str = "NUTS", zeta_ei: float = 0.01, *,
            num_mc_samples: int = 1024, acq_threshold: Optional[float] = None, verbose: bool = False) -> dict:
        """``BOBE.run`` (bo.py:967-1172): the reference's keywords in its order with its defaults (``acq='wipstd'``,
        batches of 4); keyword-only extras: ``num_mc_samples`` for the 'uniform' / 'NS' integration-point methods, an
        optional ``acq_threshold`` stop, ``verbose``.  ``acq`` may be a tuple of stages, run one after the other on the
        same surrogate (the evident intent of bo.py:1143-1156, whose tuple branch never binds ``acqs``).

        WIPV / WIPStd (``run_weighted_integrated_posterior``, bo.py:1226-1385): integration samples once before the
        loop; per iteration a kriging-believer batch, likelihood evaluations, ``update_gp``; when ``ns_n_points`` new
        evaluations have accumulated past ``min_evals`` AND the last acquisition value is <= ``logz_threshold``, nested
        sampling on the surrogate — its equal-weight samples become the next integration samples and
        (upper - lower)/2 < threshold, ``convergence_n_iters`` times in a row, ends the run ("LogZ converged",
        bo.py:886-934); otherwise fresh integration samples.  EI / LogEI (``run_EI``, bo.py:1174-1224): one point per
        iteration, log-EI goal."""
        self.min_evals, self.max_evals, self.max_gp_size = min_evals, max_evals, max_gp_size
        self.logz_threshold = logz_threshold
        self.samples_dict, self.results_dict = {}, {}
        self.convergence_n_iters = convergence_n_iters
        self.ei_goal_log = np.log(ei_goal)
        self.do_final_ns = do_final_ns
        self.fit_n_points, self.ns_n_points, self.batch_size = fit_n_points, ns_n_points, batch_size
        self.n_points_since_last_fit = 0
        self.n_points_since_last_ns = 0
        self.num_hmc_warmup, self.num_hmc_samples, self.mc_points_size = num_hmc_warmup, num_hmc_samples, mc_points_size
        self.hmc_thinning, self.hmc_num_chains, self.mc_points_method = thinning, num_chains, mc_points_method
        self.zeta_ei = zeta_ei
        self.num_mc_samples, self.acq_threshold, self.verbose = num_mc_samples, acq_threshold, verbose
        self.converged, self.convergence_counter = False, 0
        self.min_delta_seen = np.inf
        self.termination_reason = "Max evaluation budget reached"          # bo.py:1118
        self.settings.update({"min_evals": min_evals, "max_evals": max_evals, "max_gp_size": max_gp_size,
                              "logz_threshold": logz_threshold, "convergence_n_iters": convergence_n_iters,
                              "ei_goal": ei_goal, "do_final_ns": do_final_ns, "fit_n_points": fit_n_points,
                              "batch_size": batch_size, "ns_n_points": ns_n_points, "num_hmc_warmup": num_hmc_warmup,
                              "num_hmc_samples": num_hmc_samples, "mc_points_size": mc_points_size,
                              "thinning": thinning, "num_chains": num_chains, "mc_points_method": mc_points_method,
                              "zeta_ei": zeta_ei})                         # bo.py:1121-1139
        self.timing.setdefault("Nested Sampling", 0.0)
        self.timing.setdefault("MCMC Sampling", 0.0)
        self.acquisition_history = []
        self.mc_samples, self.ns_samples, self._ns_success = None, None, False
        self._current_evals = self.gp.npoints
        acqs = [acq] if isinstance(acq, str) else list(acq)
        for a in acqs:
            if a.lower() not in _ACQ:
                raise ValueError(f"Invalid acquisition function '{a}'. Valid options are: {list(_ACQ)}")
        self.current_iteration = self.start_iteration
        rs, self._resume_state = self._resume_state, None        # (a second run() on this object starts from its current state)
        # A tuple of acquisition functions runs as stages, one after the other (bo.py:1149-1158); a saved run state names the
        # stage it was written in and is continued there (the stages before it are done).
        first_stage = 0
        if rs is not None:
            k = int(rs.get("stage", 0))
            if 0 <= k < len(acqs) and str(rs.get("acq") or acqs[k]).lower() == acqs[k].lower():
                first_stage = k
            else:
                rs = None
        if rs is not None:
            # continue the interrupted run (bo.py:337-372: iteration, histories, convergence state)
            self.current_iteration, self._current_evals = int(rs["iteration"]), int(rs["current_evals"])
            self.n_points_since_last_ns, self.convergence_counter = int(rs["n_since_ns"]), int(rs["counter"])
            self.acquisition_history = list(rs["acq_history"])
            if rs.get("logz"):
                self.results_dict["logz"] = dict(rs["logz"])
            self.n_points_since_last_fit = int(rs.get("n_points_since_last_fit", 0))
            for k_, v_ in (rs.get("timing") or {}).items():
                self.timing[k_] = float(v_)
            self.np_rng.bit_generator.state = rs["rng_state"]
            self.mc_samples = rs.get("mc")
            if rs.get("converged"):                              # the saved run had already met its stopping rule
                self.converged = True
                self.termination_reason = rs.get("termination_reason", "LogZ converged")
        for k, a in enumerate(acqs):
            if k < first_stage:
                continue
            self._stage = k
            if k > first_stage:
                # every stage has its own stopping rule (the reference's loops keep `converged` local, bo.py:1188, 1257) but
                # shares the budgets: a stage that would start past them is not entered at all
                if self.check_max_evals_and_gpsize(self._current_evals):
                    break
                self.converged, self.convergence_counter = False, 0
                self.termination_reason = "Max evaluation budget reached"
            self.acquisition = _ACQ[a.lower()](optimizer=self.optimizer)
            if a.lower() == "wipv":
                self.run_WIPV(ii=self.current_iteration)
            elif a.lower() == "wipstd":
                self.run_WIPStd(ii=self.current_iteration)
            else:
                self.run_EI(ii=self.current_iteration)
        log.info(f"Final best point {self.best} with value = {self.best_f:.6f}, found at iteration {self.best_pt_iteration}")
        log.info(f"Sampling stopped: {self.termination_reason}")
        log.info(f"Final GP training set size: {self.gp.train_x.shape[0]}, max size: {self.max_gp_size}")
        self.finalise_results()
        self.start_iteration = self.current_iteration            # (a further run() on this object counts on)
        return self.results_dict

    def run_EI(self, ii=0):
        """bo.py:1174-1224: one point per iteration; stops on the (log-)EI goal or the budgets."""
        current_evals = self._current_evals
        converged = self.converged
        while not converged:
            ii += 1
            self.current_iteration = ii
            if self.verbose:
                log.info(f"Iteration {ii} of {self.acquisition.name}, objective evals {current_evals}/{self.max_evals}")
            acq_kwargs = {"zeta": self.zeta_ei,
                          "best_y": float(np.max(self.gp.train_y)) if self.gp.train_y.size > 0 else 0.0}
            n_batch = 1
            new_pts_u, acq_vals = self.get_next_batch(acq_kwargs, n_batch=n_batch, n_restarts=50, maxiter=1000,
                                                      early_stop_patience=50, step=ii, verbose=self.verbose)
            new_pts_u = np.atleast_2d(new_pts_u)
            new_vals = self.evaluate_likelihood(new_pts_u, ii, verbose=self.verbose)
            current_evals += n_batch
            self._current_evals = current_evals
            self.update_gp(new_pts_u, new_vals, step=ii, verbose=self.verbose)
            converged = self.check_convergence_ei(ii, acq_vals)
            if converged:
                self.converged = True
                self.termination_reason = f"{self.acquisition.name.upper()} goal reached"
                self.results_dict["termination_reason"] = self.termination_reason
            if self.save and ii % self.save_step == 0:
                self._checkpoint(self._run_state(), None)
            if converged:
                break
            if self.check_max_evals_and_gpsize(current_evals):
                break
        self.current_iteration = ii
        if self.save:
            self._checkpoint(self._run_state(), None)

    def _draw_mc_samples(self) -> dict:
        """The integration samples of the next iteration (bo.py:1243-1255, 1313-1324)."""
        t0 = time.time()
        if self.mc_points_method == "NUTS":
            mc = get_mc_samples(self.gp, warmup_steps=self.num_hmc_warmup, num_samples=self.num_hmc_samples,
                                thinning=self.hmc_thinning, method="NUTS", num_chains=self.hmc_num_chains,
                                np_rng=self.np_rng)
        else:
            mc = get_mc_samples(self.gp, num_samples=self.num_mc_samples, method=self.mc_points_method,
                                np_rng=self.np_rng)
        self.timing["MCMC Sampling"] += time.time() - t0
        return mc

    def run_weighted_integrated_posterior(self, acq_func_class, ii=0):
        """bo.py:1226-1390 for WIPV / WIPStd (``acq_func_class``)."""
        from .samplers import nested_sampling
        from .utils.core import resample_equal
        self.acquisition = acq_func_class(optimizer=self.optimizer)
        acq_name = self.acquisition.name
        current_evals = self._current_evals
        if self.mc_samples is None:                               # (a resumed run brings its own)
            self.mc_samples = self._draw_mc_samples()
        self.ns_samples, self._ns_success = None, False
        while not self.converged:
            ii += 1
            self.current_iteration = ii
            t_it = time.time()
            self.n_points_since_last_ns += self.batch_size
            ns_flag = self.n_points_since_last_ns >= self.ns_n_points and current_evals >= self.min_evals
            if self.verbose:
                log.info(f"Iteration {ii} of {acq_name}, objective evals {current_evals}/{self.max_evals}")
            acq_kwargs = {"mc_samples": self.mc_samples, "mc_points_size": self.mc_points_size}
            new_pts_u, acq_vals = self.get_next_batch(acq_kwargs, n_batch=self.batch_size, n_restarts=1, maxiter=100,
                                                      early_stop_patience=10, step=ii, verbose=self.verbose)  # bo.py:1274
            new_pts_u = np.atleast_2d(new_pts_u)
            acq_vals = np.atleast_1d(acq_vals)
            new_vals = self.evaluate_likelihood(new_pts_u, ii, verbose=self.verbose)
            current_evals += self.batch_size                       # a proposal the GP rejects as a duplicate still counts
            self._current_evals = current_evals
            self.update_gp(new_pts_u, new_vals, step=ii, verbose=self.verbose)
            if self.verbose:
                log.info(f"Iteration {ii}: N={self.gp.npoints} acq={self.acquisition_history[-1]:.3e} "
                         f"({time.time() - t_it:.2f} s)")
            if ns_flag and float(acq_vals[-1]) <= self.logz_threshold:        # bo.py:1283-1311
                t0 = time.time()
                ns_samples, logz_dict, ns_success = nested_sampling(self.gp, mode="convergence", dlogz=0.01,
                                                                    equal_weights=False, rng=self.np_rng)
                self.timing["Nested Sampling"] += time.time() - t0
                self.ns_samples, self._ns_success = ns_samples, ns_success
                if ns_success:
                    equal_samples, equal_logl = resample_equal(ns_samples["x"], ns_samples["logl"],
                                                               weights=ns_samples["weights"], rng=self.np_rng)   # bo.py:1296
                    self.mc_samples = {"x": equal_samples, "logl": equal_logl,
                                       "weights": np.ones(equal_samples.shape[0]), "method": "NS",
                                       "best": ns_samples["best"]}
                    self.results_dict["logz"] = logz_dict
                    self.converged = self.check_convergence_logz(ii, logz_dict, equal_samples, equal_logl,
                                                                 verbose=self.verbose)
                    if self.converged:
                        self.termination_reason = "LogZ converged"
                        self.results_dict["termination_reason"] = self.termination_reason
                self.n_points_since_last_ns = 0
            else:                                                            # bo.py:1313-1324
                self.mc_samples = self._draw_mc_samples()
            if self.acq_threshold is not None and self.acquisition_history[-1] <= self.acq_threshold \
                    and not self.converged:
                self.termination_reason = "Acquisition threshold reached"
                break
            if self.save and ii % self.save_step == 0:
                self._checkpoint(self._run_state(), self.mc_samples)
            if self.converged:
                break
            if self.check_max_evals_and_gpsize(current_evals):
                break
        self.current_iteration = ii
        if self.save:
            # the state a resumed run continues from is the state at the END OF THE LOOP: what follows (a final fit and
            # nested sampling, the result samples) draws from the generator but is not part of the iteration
            self._checkpoint(self._run_state(), self.mc_samples)
        ns_success = self._ns_success
        if self.do_final_ns and not self.converged:                          # bo.py:1345-1366
            t0 = time.time()
            gp_fit(self.gp, n_restarts=4, maxiters=500, rng=self.np_rng, factorisable_start=self.factorisable_start)
            self.timing["GP Training"] += time.time() - t0
            t0 = time.time()
            self.ns_samples, logz_dict, ns_success = nested_sampling(self.gp, mode="convergence", dlogz=0.01,
                                                                     rng=self.np_rng)
            self.timing["Nested Sampling"] += time.time() - t0
            if ns_success:
                equal_samples, equal_logl = resample_equal(self.ns_samples["x"], self.ns_samples["logl"],
                                                           weights=self.ns_samples["weights"], rng=self.np_rng)
                self.converged = self.check_convergence_logz(ii + 1, logz_dict, equal_samples, equal_logl,
                                                             verbose=self.verbose, save_checkpoint=False)
                self.results_dict["logz"] = logz_dict
                if self.converged:
                    self.termination_reason = "LogZ converged"
                    self.results_dict["termination_reason"] = self.termination_reason
            if self.save:                                                    # (the final fit changed the hyper-parameters)
                self._save_gp_file()
        if self.ns_samples is not None and ns_success:                       # bo.py:1368-1385
            x_u, weights, logl = self.ns_samples["x"], self.ns_samples["weights"], self.ns_samples["logl"]
        else:
            t0 = time.time()
            hm = get_mc_samples(self.gp, warmup_steps=512, num_samples=2000 * self.ndim, thinning=4, method="NUTS",
                                np_rng=self.np_rng)
            self.timing["MCMC Sampling"] += time.time() - t0
            x_u = hm["x"]
            weights = hm["weights"] if "weights" in hm else np.ones(hm["x"].shape[0])
            logl = hm["logp"] if "logp" in hm else hm.get("logl")
        self.samples_dict = {"x": scale_from_unit(np.asarray(x_u), self.param_bounds), "weights": np.asarray(weights),
                             "logl": np.asarray(logl)}

    def run_WIPStd(self, ii=0):
        """bo.py:1392-1394."""
        return self.run_weighted_integrated_posterior(WIPStd, ii)

    def run_WIPV(self, ii=0):
        """bo.py:1396-1398."""
        return self.run_weighted_integrated_posterior(WIPV, ii)
