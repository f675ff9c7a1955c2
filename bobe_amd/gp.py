"""``GP`` — MI355X-resident counterpart of ``BOBE.gp.GP`` (BOBE/gp.py:199-772).

Same constructor keywords, attributes and methods as the reference class; all O(N^2)/O(N^3)
arithmetic (kernel assembly, Cholesky, solves, MLL + gradient, batched prediction, fantasy
variance) runs on the GPU through libbobe_gp.so (include/bobe_gp.h).  Host side keeps only what
the reference also keeps on the host: y standardisation, priors, bounds, the L-BFGS-B loop.

There is no CPU fallback: constructing a GP without the library or without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import List, Optional

import numpy as np

from . import _lib
from . import priors as P
from .optim import optimize_optax, optimize_scipy
from .utils import get_logger, get_numpy_rng

log = get_logger("gp")

safe_noise_floor = 1e-12  # BOBE/gp.py:16

KERNEL_IDS = {"rbf": 0, "matern": 1}


def _tofloat(x) -> float:
    return float(np.asarray(x).reshape(-1)[0]) if np.ndim(x) else float(x)


# ---- module-level helpers of the reference's gp.py, same names and arguments --------------------------------
DummyDistribution = P.Dummy                      # gp.py:22-25
make_distribution = P.make_distribution          # gp.py:27-55 (numpyro's names and keywords, priors.py)


def saas_prior_logprob(lengthscales, kernel_variance, tausq):
    """gp.py:57-78: the SAAS prior's log-density (O(d) host arithmetic, priors.py)."""
    return P.saas_logprob_and_grad(np.asarray(lengthscales, dtype=np.float64), float(kernel_variance), float(tausq))[0]


_KERNEL_HANDLES: dict = {}


_FREE_LOCK = threading.RLock()     # the module-level functions share handles: one caller at a time


def _free_handle(kernel_id: int, d: int, device: int = 0):
    """The cached data-less handle per (kernel, d, device) behind the module-level kernel / distance functions (they hold
    O(n1 n2) scratch only).  Callers hold ``_FREE_LOCK`` for the duration of the library call."""
    lib = _lib.load()
    key = (kernel_id, d, device)
    if key not in _KERNEL_HANDLES:
        h = C.c_void_p(0)
        _lib.check(lib.bobe_gp_create(C.byref(h), device, kernel_id, d), "bobe_gp_create")
        _KERNEL_HANDLES[key] = h
    return lib, _KERNEL_HANDLES[key]


class _scratch_handle:
    """A data-less handle for ONE call of a matrix-input function (``gp_mll``, ``fast_update_cholesky``): the library sizes a
    factorisation workspace of several n^2 doubles for it, which goes back to the device when the call is over."""

    def __init__(self, device: int = 0):
        self.device = device

    def __enter__(self):
        self.lib = _lib.load()
        self.h = C.c_void_p(0)
        _lib.check(self.lib.bobe_gp_create(C.byref(self.h), self.device, KERNEL_IDS["rbf"], 1), "bobe_gp_create")
        return self.lib, self.h

    def __exit__(self, *exc):
        self.lib.bobe_gp_destroy(self.h)
        return False


def dist_sq(x, y):
    """gp.py:80-96: squared Euclidean distances (n1, n2) of the rows of ``x`` and ``y``, on the GPU (bobe_gp_dist_sq)."""
    x = _lib.as_f64(np.atleast_2d(x))
    y = _lib.as_f64(np.atleast_2d(y))
    if x.shape[1] != y.shape[1]:
        raise ValueError("x and y must have the same number of columns")
    out = np.empty((x.shape[0], y.shape[0]))
    with _FREE_LOCK:
        lib, h = _free_handle(KERNEL_IDS["rbf"], x.shape[1])
        _lib.check(lib.bobe_gp_dist_sq(h, _lib.ptr(x), x.shape[0], _lib.ptr(y), y.shape[0], _lib.ptr(out)), "bobe_gp_dist_sq")
    return out


def gp_mll(k, train_y, num_points):
    """gp.py:170-178: the marginal log-likelihood of ``train_y`` under the kernel matrix ``k`` handed in (Cholesky, solves
    and reductions on the GPU, bobe_gp_mll_from_k); NaN when ``k`` is not positive definite, as XLA's Cholesky gives."""
    k = _lib.as_f64(k)
    y = _lib.as_f64(train_y).reshape(-1)
    n = int(num_points)
    if k.shape != (n, n) or y.shape[0] != n:
        raise ValueError(f"k must be ({n}, {n}) and train_y must hold {n} values")
    mll = C.c_double(0.0)
    with _scratch_handle() as (lib, h):
        _lib.check(lib.bobe_gp_mll_from_k(h, _lib.ptr(k), n, _lib.ptr(y), C.byref(mll)), "bobe_gp_mll_from_k")
    return float(mll.value)


def fast_update_cholesky(L, k, k_self):
    """gp.py:181-197: the (n+1) x (n+1) factor after appending one point - last row ``[L^-1 k, sqrt(k_self - |L^-1 k|^2)]``,
    the solve on the GPU (bobe_gp_chol_row_update)."""
    L = _lib.as_f64(L)
    n = L.shape[0]
    k = _lib.as_f64(k).reshape(-1)
    if L.shape != (n, n) or k.shape[0] != n:
        raise ValueError("L must be (n, n) and k must hold n values")
    v = np.empty(n)
    diag = C.c_double(0.0)
    with _scratch_handle() as (lib, h):
        _lib.check(lib.bobe_gp_chol_row_update(h, _lib.ptr(L), n, _lib.ptr(k), _tofloat(k_self), _lib.ptr(v), C.byref(diag)),
                   "bobe_gp_chol_row_update")
    new_L = np.zeros((n + 1, n + 1))
    new_L[:n, :n] = L
    new_L[n, :n] = v
    new_L[n, n] = diag.value
    return new_L


def _kernel_on_gpu(kernel_id: int, xa, xb, lengthscales, kernel_variance, noise, include_noise, device: int = 0):
    """K(xa, xb) through bobe_gp_kernel on a cached data-less handle per (kernel, d, device)."""
    xa = _lib.as_f64(np.atleast_2d(xa))
    xb = _lib.as_f64(np.atleast_2d(xb))
    d = xa.shape[1]
    ls = _lib.as_f64(lengthscales).reshape(-1)
    if ls.size == 1 and d > 1:
        ls = np.full(d, float(ls[0]))
    out = np.empty((xa.shape[0], xb.shape[0]))
    with _FREE_LOCK:
        lib, handle = _free_handle(kernel_id, d, device)
        _lib.check(lib.bobe_gp_kernel(handle, _lib.ptr(xa), xa.shape[0], _lib.ptr(xb), xb.shape[0],
                                      _lib.ptr(ls), float(kernel_variance), float(noise), 1 if include_noise else 0,
                                      _lib.ptr(out)), "bobe_gp_kernel")
    return out


def rbf_kernel(xa, xb, lengthscales, kernel_variance, noise, include_noise=True):
    """BOBE/gp.py:124-154, evaluated on the GPU."""
    return _kernel_on_gpu(KERNEL_IDS["rbf"], xa, xb, lengthscales, kernel_variance, noise, include_noise)


def matern_kernel(xa, xb, lengthscales, kernel_variance, noise, include_noise=True):
    """BOBE/gp.py:156-168 (Matérn-5/2), evaluated on the GPU."""
    return _kernel_on_gpu(KERNEL_IDS["matern"], xa, xb, lengthscales, kernel_variance, noise, include_noise)


def kernel_diag(x, kernel_variance, noise, include_noise=True):
    """BOBE/gp.py:98-122: the constant diagonal of a stationary kernel (O(n) host vector)."""
    diag = float(kernel_variance) * np.ones(np.atleast_2d(x).shape[0])
    return diag + float(noise) if include_noise else diag


class GP:
    def __init__(self, train_x, train_y, noise=1e-8, kernel="rbf", optimizer="scipy", optimizer_options={},
                 kernel_variance_bounds=[1e-4, 1e8], lengthscale_bounds=[0.01, 5], lengthscales=None,
                 kernel_variance=None, kernel_variance_prior=None, lengthscale_prior=None, tausq=None,
                 tausq_bounds=[1e-4, 1e4], param_names: Optional[List[str]] = None, device: int = 0,
                 pivot_floor_ulp: Optional[float] = None, _factor: bool = True):
        """Same keywords as BOBE/gp.py:201-203 plus ``device`` (HIP device index) and ``pivot_floor_ulp`` (None: the
        library's default, 0 = the reference's rule - a factorisation fails only on a pivot <= 0, gp.py:175, 549; 64 is
        what ``BOBE(...)`` passes, see the property).  ``_factor=False`` (internal) leaves the factorisation to the caller,
        which is about to install a known one (``from_state_dict``, ``copy``)."""
        self._lib = _lib.load()
        self._h = C.c_void_p(0)
        self.device = int(device)
        self._setup_training_data(train_x, train_y)
        self.param_names = param_names if param_names is not None else ["x_" + str(i) for i in range(self.ndim)]

        self.kernel_name = kernel if kernel == "rbf" else "matern"              # gp.py:251
        self.lengthscales = (np.array(lengthscales, dtype=np.float64).reshape(-1) if lengthscales is not None
                             else np.ones(self.ndim))
        self.kernel_variance = float(kernel_variance) if kernel_variance is not None else 1.0
        self.noise = float(noise)

        _lib.check(self._lib.bobe_gp_create(C.byref(self._h), self.device, KERNEL_IDS[self.kernel_name], self.ndim),
                   "bobe_gp_create")

        self.optimizer_method = optimizer
        self.mll_optimize = optimize_scipy if optimizer == "scipy" else optimize_optax          # gp.py:264-267
        self.optimizer_options = optimizer_options
        self.concurrent_restarts = True        # fit(): the restarts run concurrently ...
        self.restart_slots = 4                 # ... this many evaluations in flight at once (more oversubscribes the queues)
        # ... advancing in lock step, one bobe_gp_mll_batch call per round, the L-BFGS-B routines stepped by one thread
        # ("lockstep"), or each in its own thread on an evaluation slot ("slots").  Same trajectories and result either
        # way; "auto" = lock step where SciPy's routine can be stepped (optim._rc_available), else slots: measured
        # 33.7 vs 53.5 ms at N = 100, 39.5 vs 45.7 at 600, 124 vs 141 at 2048, 457 vs 461 at 4096 for a 4-restart fit
        self.restart_mode = "auto"

        self.lengthscale_bounds = lengthscale_bounds
        self.kernel_variance_bounds = kernel_variance_bounds
        self.tausq = float(tausq) if tausq is not None else 1.0
        self.tausq_bounds = tausq_bounds

        self._setup_kernel_variance_prior(kernel_variance_prior)
        self._setup_lengthscale_prior(lengthscale_prior)
        self._setup_optimization_parameters()

        self.append_updates = True             # update(): rank-b append (O(b N^2)) instead of a full refactorisation
        self._chol_cache = self._alpha_cache = None
        self._pushed_hyper = None              # hyper-parameters of the factor on the device (set by _push_hyper)
        self.not_pd = False
        self._rank_test_noted = False
        if pivot_floor_ulp is not None:
            self.pivot_floor_ulp = pivot_floor_ulp
        self._push_data()
        if _factor:
            self.recompute_cholesky()                                           # gp.py:257-260

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.bobe_gp_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass

    # ------------------------------------------------------------------ data
    def _setup_training_data(self, train_x, train_y):
        """BOBE/gp.py:283-307."""
        train_x = np.asarray(train_x, dtype=np.float64)
        train_y = np.asarray(train_y, dtype=np.float64)
        if train_x.shape[0] != train_y.shape[0]:
            raise ValueError("train_x and train_y must have the same number of points")
        if train_y.ndim != 2:
            train_y = train_y.reshape(-1, 1)
        if train_x.ndim != 2:
            raise ValueError("train_x must be 2D")
        self.ndim = train_x.shape[1]
        self.y_mean = float(np.mean(train_y)) if train_y.size > 0 else 0.0
        self.y_std = float(np.std(train_y)) if train_y.size > 0 else 1.0
        if self.y_std == 0:
            log.warning("Training targets have zero variance. Setting std to 1.0 to avoid division by zero.")
            self.y_std = 1.0
        self.train_x = np.array(train_x)
        self.train_y = (train_y - self.y_mean) / self.y_std

    def _push_data(self):
        x = _lib.as_f64(self.train_x)
        y = _lib.as_f64(self.train_y).reshape(-1)
        _lib.check(self._lib.bobe_gp_set_data(self._h, _lib.ptr(x), _lib.ptr(y), x.shape[0]), "bobe_gp_set_data")

    def _push_hyper(self):
        ls = _lib.as_f64(self.lengthscales).reshape(-1)
        _lib.check(self._lib.bobe_gp_set_hyper(self._h, _lib.ptr(ls), float(self.kernel_variance), float(self.noise)),
                   "bobe_gp_set_hyper")
        self._pushed_hyper = self._hyper_key()

    def _hyper_key(self):
        """The hyper-parameters as the device holds them after a ``_push_hyper`` (what the current factor was built with)."""
        return (tuple(np.asarray(self.lengthscales, dtype=np.float64).reshape(-1).tolist()), float(self.kernel_variance),
                float(self.noise))

    # ------------------------------------------------------------------ priors / bounds
    def _setup_kernel_variance_prior(self, kernel_variance_prior):
        """BOBE/gp.py:309-320."""
        self.kernel_variance_prior_spec = kernel_variance_prior
        if self.kernel_variance_prior_spec is None:
            self.kernel_variance_prior_spec = {"name": "Uniform", "low": self.kernel_variance_bounds[0],
                                               "high": self.kernel_variance_bounds[1]}
        self.fixed_kernel_variance = (self.kernel_variance_prior_spec == "fixed")
        self.kernel_variance_prior_dist = (P.Dummy() if self.fixed_kernel_variance
                                           else P.make_distribution(self.kernel_variance_prior_spec))

    def _setup_lengthscale_prior(self, lengthscale_prior):
        """BOBE/gp.py:322-337."""
        self.lengthscale_prior_spec = lengthscale_prior
        if self.lengthscale_prior_spec is None:
            self.lengthscale_prior_spec = {"name": "Uniform", "low": self.lengthscale_bounds[0],
                                           "high": self.lengthscale_bounds[1]}
        if self.lengthscale_prior_spec == "DSLP":
            self.lengthscale_prior_dist = P.dslp(self.ndim)
        elif self.lengthscale_prior_spec == "SAAS":
            self.lengthscale_prior_dist = None
        else:
            self.lengthscale_prior_dist = P.make_distribution(self.lengthscale_prior_spec)

    def _setup_optimization_parameters(self):
        """BOBE/gp.py:339-355."""
        self.hyperparam_names = ["lengthscales"]
        bounds = [self.lengthscale_bounds] * self.ndim
        if not self.fixed_kernel_variance:
            self.hyperparam_names.append("kernel_variance")
            bounds.append(self.kernel_variance_bounds)
        if self.lengthscale_prior_spec == "SAAS":
            self.hyperparam_names.append("tausq")
            bounds.append(self.tausq_bounds)
        self.hyperparam_bounds = np.log(np.array(bounds, dtype=np.float64).T)
        self.num_hyperparams = self.hyperparam_bounds.shape[1]

    def prior_func(self, lengthscales, kernel_variance, tausq=None) -> float:
        """BOBE/gp.py:357-366."""
        return self._prior_and_grad(lengthscales, kernel_variance, tausq)[0]

    def _prior_and_grad(self, lengthscales, kernel_variance, tausq):
        """log prior and its derivative wrt (ls, kvar, tausq) (not yet in log space)."""
        ls = np.asarray(lengthscales, dtype=np.float64)
        if self.lengthscale_prior_spec == "SAAS":
            return P.saas_logprob_and_grad(ls, kernel_variance, tausq)
        # the reference's default - Uniform on both (gp.py:313, 326) - is a constant with zero gradient: evaluated once (a
        # fit asks thousands of times), by the same expressions as below so that the value keeps its bits
        flat = (type(self.kernel_variance_prior_dist), type(self.lengthscale_prior_dist)) == (P.Uniform, P.Uniform)
        key = flat and (self.kernel_variance_prior_dist.low, self.kernel_variance_prior_dist.high,
                        self.lengthscale_prior_dist.low, self.lengthscale_prior_dist.high, ls.shape)
        if flat and getattr(self, "_flat_prior_key", None) == key:
            return self._flat_prior
        lp = float(np.sum(self.kernel_variance_prior_dist.log_prob(kernel_variance)))
        g_kvar = float(np.sum(self.kernel_variance_prior_dist.dlog_prob(kernel_variance)))
        lp += float(np.sum(self.lengthscale_prior_dist.log_prob(ls)))
        g_ls = np.asarray(self.lengthscale_prior_dist.dlog_prob(ls), dtype=np.float64)
        if flat:
            g_ls.setflags(write=False)
            self._flat_prior_key, self._flat_prior = key, (lp, g_ls, g_kvar, 0.0)
        return lp, g_ls, g_kvar, 0.0

    def _parse_hyperparams(self, log_params):
        """BOBE/gp.py:368-383."""
        hp = np.exp(np.asarray(log_params, dtype=np.float64))
        ls = hp[:self.ndim]
        if self.fixed_kernel_variance:
            kvar = self.kernel_variance
            if "tausq" in self.hyperparam_names:
                tausq = hp[self.ndim] if len(hp) > self.ndim else self.tausq
            else:
                tausq = self.tausq
        else:
            kvar = hp[self.ndim]
            tausq = hp[self.ndim + 1] if len(hp) > self.ndim + 1 else self.tausq
        return ls, float(kvar), float(tausq)

    # ------------------------------------------------------------------ objective
    def mll_data(self, lengthscales, kernel_variance, want_grad=True, slot=None):
        """Data term of the MLL (gp_mll, gp.py:170-178) and its gradient wrt (log ls, log kvar) on the GPU.
        ``slot`` (0..7): evaluate on that private stream / workspace (bobe_gp_mll_submit + _wait) so that several
        host threads — the restarts of ``fit`` — can have evaluations in flight together; same bits either way."""
        ls = _lib.as_f64(lengthscales).reshape(-1)
        mll = C.c_double(0.0)
        grad = np.empty(self.ndim + 1) if want_grad else None
        if slot is None:
            st = self._lib.bobe_gp_mll(self._h, _lib.ptr(ls), float(kernel_variance), C.byref(mll), _lib.ptr(grad))
            _lib.check(st, "bobe_gp_mll")
        else:
            _lib.check(self._lib.bobe_gp_mll_submit(self._h, int(slot), _lib.ptr(ls), float(kernel_variance),
                                                    int(want_grad)), "bobe_gp_mll_submit")
            _lib.check(self._lib.bobe_gp_mll_wait(self._h, int(slot), C.byref(mll), _lib.ptr(grad)), "bobe_gp_mll_wait")
        return mll.value, grad

    def mll_data_batch(self, lengthscales, kernel_variances, want_grad=True):
        """``mll_data`` for B hyper-parameter vectors evaluated concurrently on the GPU (bobe_gp_mll_batch).
        Vectors whose kernel matrix is not positive definite come back as NaN, like ``mll_data``."""
        ls = _lib.as_f64(lengthscales).reshape(-1, self.ndim)
        B = ls.shape[0]
        kv = _lib.as_f64(kernel_variances).reshape(B)
        mll = np.empty(B)
        grad = np.empty((B, self.ndim + 1)) if want_grad else None
        status = np.zeros(B, dtype=np.int32)
        st = self._lib.bobe_gp_mll_batch(self._h, B, _lib.ptr(ls), _lib.ptr(kv), _lib.ptr(mll), _lib.ptr(grad),
                                         C.c_void_p(status.ctypes.data))
        _lib.check(st, "bobe_gp_mll_batch")
        return mll, grad

    def neg_mll(self, log_params):
        """BOBE/gp.py:385-398."""
        return self.neg_mll_value_and_grad(log_params, want_grad=False)[0]

    def neg_mll_value_and_grad(self, log_params, want_grad=True, slot=None):
        """(f, df/dtheta) with f = -(MLL + log prior), theta = log hp — the closure optim.py:306-309 builds."""
        log_params = np.asarray(log_params, dtype=np.float64)
        ls, kvar, tausq = self._parse_hyperparams(log_params)
        mll, g_data = self.mll_data(ls, kvar, want_grad, slot=slot)
        return self._assemble_objective(log_params, ls, kvar, tausq, mll, g_data, want_grad)

    def neg_mll_value_and_grad_batch(self, log_params_list, want_grad=True):
        """``neg_mll_value_and_grad`` for several theta at once (the concurrently running restarts of ``fit``);
        returns a list of (f, grad) pairs with exactly the values of the one-at-a-time call."""
        thetas = [np.asarray(t, dtype=np.float64) for t in log_params_list]
        parsed = [self._parse_hyperparams(t) for t in thetas]
        mll, g_data = self.mll_data_batch(np.array([p[0] for p in parsed]), np.array([p[1] for p in parsed]), want_grad)
        return [self._assemble_objective(t, p[0], p[1], p[2], float(mll[i]), None if g_data is None else g_data[i],
                                         want_grad) for i, (t, p) in enumerate(zip(thetas, parsed))]

    def _assemble_objective(self, log_params, ls, kvar, tausq, mll, g_data, want_grad):
        lp, g_ls, g_kvar, g_tau = self._prior_and_grad(ls, kvar, tausq)
        val = -(mll + lp)
        if not want_grad:
            return val, None
        grad = np.zeros(len(log_params))
        grad[:self.ndim] = g_data[:self.ndim] + g_ls * ls                     # chain rule: d/dlog x = x d/dx
        idx = self.ndim
        if not self.fixed_kernel_variance:
            grad[idx] = g_data[self.ndim] + g_kvar * kvar
            idx += 1
        if "tausq" in self.hyperparam_names and len(log_params) > idx:
            grad[idx] = g_tau * tausq
        return val, -grad

    def fit(self, x0: np.ndarray = None, maxiter: int = 500) -> dict:
        """BOBE/gp.py:400-437."""
        if x0 is None:
            x0 = np.log(self.get_hyperparams())[None, :]
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        optimizer_options = dict(self.optimizer_options)
        # restarts are independent L-BFGS-B runs: each gets a host thread and an evaluation slot of the library
        extra = {}
        if self.concurrent_restarts and x0.shape[0] > 1 and self.mll_optimize is optimize_scipy:
            from .optim import _rc_available
            if self.restart_mode == "lockstep" or (self.restart_mode == "auto" and _rc_available()):
                extra = {"batch_value_and_grad": self.neg_mll_value_and_grad_batch}
            else:
                extra = {"slot_value_and_grad": lambda x, slot: self.neg_mll_value_and_grad(x, slot=slot),
                         "n_slots": self.restart_slots}
        best_params_log, best_loss = self.mll_optimize(
            self.neg_mll_value_and_grad, num_params=self.num_hyperparams, bounds=self.hyperparam_bounds, x0=x0,
            maxiter=maxiter, n_restarts=x0.shape[0], optimizer_options=optimizer_options, **extra)
        return {"mll": -best_loss, "params": best_params_log}

    def update_hyperparams(self, hyperparams):
        """BOBE/gp.py:439-448."""
        ls, kvar, tausq = self._parse_hyperparams(hyperparams)
        self.lengthscales = np.array(ls)
        if not self.fixed_kernel_variance:
            self.kernel_variance = kvar
        self.tausq = tausq
        self.recompute_cholesky()

    def recompute_cholesky(self):
        """BOBE/gp.py:544-550 — K, L, alpha on the GPU.  Not-PD leaves NaNs, like XLA."""
        self._push_hyper()
        st = _lib.check(self._lib.bobe_gp_factor(self._h), "bobe_gp_factor")
        self._chol_cache = None
        self._alpha_cache = None
        self.not_pd = (st == _lib.BOBE_NOT_PD)
        if self.not_pd:
            self._note_rank_test()

    def _note_rank_test(self) -> None:
        """Logged once per GP: a factorisation was refused by the rank test (a positive pivot below ``pivot_floor_ulp``
        machine epsilons of k(x,x) + noise), i.e. where LAPACK's sign test - the reference's rule - may have passed."""
        if not self._rank_test_noted and "rank test" in _lib.last_error():
            self._rank_test_noted = True
            log.warning(f"{_lib.last_error()} at kernel variance {self.kernel_variance:.3g}, noise {self.noise:.3g}: "
                        "treated as not positive definite (NaN).  pivot_floor_ulp = 0 (GP attribute, or "
                        "BOBE(gp_kwargs={'pivot_floor_ulp': 0})) restores the reference's sign-only test.")

    @property
    def refine_kappa(self) -> float:
        """Where (kernel_variance + noise) / smallest pivot of the factor exceeds this, v = L^-1 k is solved for by blocked
        forward substitution (the reference's ``solve_triangular``, gp.py:462, 484, 571) instead of multiplied out with the
        inverse factor (include/bobe_gp.h, bobe_gp_set_refine_kappa).  1e6 by default, 0 = always, negative = never; takes
        effect at the next factorisation."""
        k = C.c_double()
        _lib.check(self._lib.bobe_gp_get_refine(self._h, C.byref(k), None), "bobe_gp_get_refine")
        return float(k.value)

    @refine_kappa.setter
    def refine_kappa(self, kappa: float) -> None:
        _lib.check(self._lib.bobe_gp_set_refine_kappa(self._h, float(kappa)), "bobe_gp_set_refine_kappa")

    @property
    def solve_block(self) -> int:
        """Rows of the diagonal blocks of that substitution (a positive multiple of 128, default 128: the accuracy of
        LAPACK's triangular solve or better; include/bobe_gp.h, bobe_gp_set_solve_block)."""
        return int(self._lib.bobe_gp_get_solve_block(self._h))

    @solve_block.setter
    def solve_block(self, rows: int) -> None:
        _lib.check(self._lib.bobe_gp_set_solve_block(self._h, int(rows)), "bobe_gp_set_solve_block")

    @property
    def refining(self) -> bool:
        """Whether the current factor counts as ill conditioned: v = L^-1 k is solved for, not multiplied out."""
        a = C.c_int()
        _lib.check(self._lib.bobe_gp_get_refine(self._h, None, C.byref(a)), "bobe_gp_get_refine")
        return bool(a.value)

    @property
    def pivot_floor_ulp(self) -> float:
        """The rank test's factor (include/bobe_gp.h, "Conventions"): 0 by default = the reference's rule (only a pivot <= 0
        fails, as LAPACK's dpotrf reports it, gp.py:175, 549); u > 0: a positive pivot below u machine epsilons of k(x,x) +
        noise fails too (``BOBE(...)`` runs its surrogate with 64).  Takes effect at the next factorisation / MLL
        evaluation."""
        return float(self._lib.bobe_gp_get_pivot_floor_ulp(self._h))

    @pivot_floor_ulp.setter
    def pivot_floor_ulp(self, ulp: float) -> None:
        _lib.check(self._lib.bobe_gp_set_pivot_floor_ulp(self._h, float(ulp)), "bobe_gp_set_pivot_floor_ulp")

    # ------------------------------------------------------------------ state on the host (lazy)
    @property
    def cholesky(self) -> np.ndarray:
        """GP.cholesky (gp.py:259): N x N lower factor, fetched from the GPU on demand."""
        if self._chol_cache is None:
            n = self.npoints
            L = np.empty((n, n))
            _lib.check(self._lib.bobe_gp_get_chol(self._h, _lib.ptr(L), None), "bobe_gp_get_chol")
            self._chol_cache = L
        return self._chol_cache

    @property
    def alphas(self) -> np.ndarray:
        """GP.alphas (gp.py:260): (N, 1)."""
        if self._alpha_cache is None:
            a = np.empty(self.npoints)
            _lib.check(self._lib.bobe_gp_get_chol(self._h, None, _lib.ptr(a)), "bobe_gp_get_chol")
            self._alpha_cache = a.reshape(-1, 1)
        return self._alpha_cache

    # ------------------------------------------------------------------ kernel
    def kernel(self, xa, xb, lengthscales=None, kernel_variance=None, noise=None, include_noise=True):
        """GP.kernel(xa, xb, ls, kvar, noise, include_noise) (gp.py:124-168; call site acquisition.py:388).

        Evaluated on the GPU with the hyper-parameters passed (defaults: the GP's own)."""
        xa = _lib.as_f64(np.atleast_2d(xa))
        xb = _lib.as_f64(np.atleast_2d(xb))
        ls = _lib.as_f64(lengthscales if lengthscales is not None else self.lengthscales).reshape(-1)
        kvar = float(kernel_variance if kernel_variance is not None else self.kernel_variance)
        nz = float(noise if noise is not None else self.noise)
        out = np.empty((xa.shape[0], xb.shape[0]))
        _lib.check(self._lib.bobe_gp_kernel(self._h, _lib.ptr(xa), xa.shape[0], _lib.ptr(xb), xb.shape[0],
                                            _lib.ptr(ls), kvar, nz, 1 if include_noise else 0, _lib.ptr(out)),
                   "bobe_gp_kernel")
        return out

    # ------------------------------------------------------------------ prediction
    def _predict(self, x, want_mean, want_var, policy):
        x = _lib.as_f64(np.atleast_2d(x))
        c = x.shape[0]
        mean = np.empty(c) if want_mean else None
        var = np.empty(c) if want_var else None
        _lib.check(self._lib.bobe_gp_predict(self._h, _lib.ptr(x), c, _lib.ptr(mean), _lib.ptr(var), policy),
                   "bobe_gp_predict")
        return mean, var

    def predict_mean_batched(self, x):
        """BOBE/gp.py:450-457, 468-470."""
        m, _ = self._predict(x, True, False, 0)
        return m * self.y_std + self.y_mean

    def predict_mean_single(self, x):
        return self.predict_mean_batched(x)[0]

    def predict_var_batched(self, x):
        """BOBE/gp.py:459-466, 472-474 (clip keeps NaN)."""
        _, v = self._predict(x, False, True, 0)
        return self.y_std ** 2 * v

    def predict_var_single(self, x):
        return self.predict_var_batched(x)[0]

    def predict_batched(self, x):
        """BOBE/gp.py:476-493 — standardised (mean, var), NaN / < 1e-12 -> 1e-12."""
        return self._predict(x, True, True, 1)

    def predict_single(self, x):
        m, v = self.predict_batched(x)
        return m[0], v[0:1]

    # ------------------------------------------------------------------ update
    def update(self, new_x, new_y):
        """BOBE/gp.py:495-541 — duplicate filter, re-standardise, full refactor."""
        new_x = np.atleast_2d(np.asarray(new_x, dtype=np.float64))
        new_y = np.atleast_2d(np.asarray(new_y, dtype=np.float64))
        pts, vals = [], []
        for i in range(new_x.shape[0]):
            if np.any(np.all(np.isclose(self.train_x, new_x[i], atol=1e-6, rtol=1e-4), axis=1)):
                log.debug(f"Point {new_x[i]} already exists in the training set, not updating")
            else:
                pts.append(new_x[i])
                vals.append(new_y[i])
        if pts:
            self.train_x = np.vstack([self.train_x, np.array(pts)])
            y_orig = np.vstack([self.train_y * self.y_std + self.y_mean, np.array(vals).reshape(-1, 1)])
            self.y_mean = float(np.mean(y_orig))
            self.y_std = float(np.std(y_orig))
            if self.y_std == 0:
                log.warning("Training targets have zero variance. Setting std to 1.0 to avoid division by zero.")
                self.y_std = 1.0
            self.train_y = (y_orig - self.y_mean) / self.y_std
            self._refresh_after_update(len(pts))

    def _refresh_after_update(self, n_new: int):
        """K, L, alpha for the grown training set (gp.py:541).  While the hyper-parameters are the ones the factor on
        the device was built with, the factor of the old points is still valid: the new rows are appended on the GPU
        (``bobe_gp_append``, the b-row form of the reference's own ``fast_update_cholesky``, gp.py:181-197) and alpha
        is re-solved for the re-standardised targets — O(b N^2) instead of O(N^3).  Everything else takes the
        reference's route, ``recompute_cholesky()`` from the CURRENT attributes (gp.py:541-550, "useful if
        hyperparameters are changed manually"): lengthscales / kernel_variance / noise set by hand since the last
        factorisation, no usable factor (NaN state), a large batch, or an append that failed."""
        if (self.append_updates and not self.not_pd and 1 <= n_new <= 64 and self.train_x.shape[0] > n_new
                and self._pushed_hyper == self._hyper_key()):
            try:
                st = self._append_rows(n_new)
            except _lib.BobeLibraryError as e:
                # (the library leaves a failed append as an empty handle: rebuild it from the host copy of the data)
                log.warning(f"append of {n_new} rows failed ({e}); refactorising")
            else:
                self._chol_cache = self._alpha_cache = None
                self.not_pd = (st == _lib.BOBE_NOT_PD)
                return
        self._push_data()
        self.recompute_cholesky()

    def _append_rows(self, n_new: int) -> int:
        xn = _lib.as_f64(self.train_x[-n_new:])
        ya = _lib.as_f64(self.train_y).reshape(-1)
        return _lib.check(self._lib.bobe_gp_append(self._h, _lib.ptr(xn), n_new, _lib.ptr(ya)), "bobe_gp_append")

    # ------------------------------------------------------------------ fantasy variance / sweep
    def fantasy_var(self, new_x, mc_points, k_train_mc=None):
        """BOBE/gp.py:552-576.  ``k_train_mc`` is accepted for signature parity and recomputed on the GPU.
        ``new_x`` may hold several candidates; the result is then (C, M)."""
        new_x = _lib.as_f64(np.atleast_2d(new_x))
        z = _lib.as_f64(np.atleast_2d(mc_points))
        out = np.empty((new_x.shape[0], z.shape[0]))
        _lib.check(self._lib.bobe_gp_fantasy_var(self._h, _lib.ptr(new_x), new_x.shape[0], _lib.ptr(z), z.shape[0],
                                                 float(self.y_std), _lib.ptr(out)), "bobe_gp_fantasy_var")
        return out[0] if out.shape[0] == 1 else out

    def wip_sweep(self, candidates, mc_points, want_mean_var=False):
        """All-candidate WIPV / WIPStd scores (acquisition.py:385-398 + 438-465) in one GPU call.

        ``candidates`` / ``mc_points`` may be NumPy arrays or torch CUDA tensors (no copy then).
        Returns dict(wipv, wipstd, argmin_v, min_v, argmin_s, min_s[, mean, var])."""
        dev = hasattr(candidates, "data_ptr")
        cand = candidates if dev else _lib.as_f64(np.atleast_2d(candidates))
        z = mc_points if hasattr(mc_points, "data_ptr") else _lib.as_f64(np.atleast_2d(mc_points))
        c, m = int(cand.shape[0]), int(z.shape[0])
        wipv, wipstd = np.empty(c), np.empty(c)
        mean = np.empty(c) if want_mean_var else None
        var = np.empty(c) if want_mean_var else None
        av, asd = C.c_int64(-1), C.c_int64(-1)
        mv, ms = C.c_double(0.0), C.c_double(0.0)
        _lib.check(self._lib.bobe_gp_wip_sweep(self._h, _lib.ptr(cand), c, _lib.ptr(z), m, float(self.y_std),
                                               _lib.ptr(wipv), _lib.ptr(wipstd), _lib.ptr(mean), _lib.ptr(var),
                                               C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)),
                   "bobe_gp_wip_sweep")
        out = {"wipv": wipv, "wipstd": wipstd, "argmin_v": av.value, "min_v": mv.value, "argmin_s": asd.value,
               "min_s": ms.value}
        if want_mean_var:
            out["mean"], out["var"] = mean, var
        return out

    def acq_ei(self, x, best_y, zeta=0.0, log_ei=False):
        """+EI / +log EI at x (acquisition.py:226-253, 318-330), standardised units."""
        x = _lib.as_f64(np.atleast_2d(x))
        out = np.empty(x.shape[0])
        _lib.check(self._lib.bobe_gp_acq_ei(self._h, _lib.ptr(x), x.shape[0], float(best_y), float(zeta),
                                            1 if log_ei else 0, _lib.ptr(out)), "bobe_gp_acq_ei")
        return out

    def wip_grad(self, candidates, mc_points):
        """(wipv, wipstd, dwipv/dx, dwipstd/dx) of the candidates — values and input gradients of ``WIPV.fun`` /
        ``WIPStd.fun`` (acquisition.py:438-465), the gradient being what the reference takes with ``jax.grad`` in the
        local refinement (acquisition.py:403-412).  Physical units (the y_std factors of gp.py:576 included)."""
        cand = _lib.as_f64(np.atleast_2d(candidates))
        z = _lib.as_f64(np.atleast_2d(mc_points))
        c = cand.shape[0]
        wipv, wipstd = np.empty(c), np.empty(c)
        dv, ds = np.empty((c, self.ndim)), np.empty((c, self.ndim))
        _lib.check(self._lib.bobe_gp_wip_grad(self._h, _lib.ptr(cand), c, _lib.ptr(z), z.shape[0], float(self.y_std),
                                              _lib.ptr(wipv), _lib.ptr(wipstd), _lib.ptr(dv), _lib.ptr(ds)),
                   "bobe_gp_wip_grad")
        return wipv, wipstd, dv, ds

    def predict_grad(self, x, mean_only=False):
        """Standardised (mean, var, dmean/dx, dvar/dx) of ``predict_single`` (gp.py:476-489) for C points: what the
        reference gets by JAX autodiff through the GP (acquisition.py:246-253, samplers.py:268-276).
        ``mean_only``: (mean, None, dmean/dx, None) from one small kernel — the HMC sampler's call."""
        x = _lib.as_f64(np.atleast_2d(x))
        c = x.shape[0]
        if mean_only:
            mean, dmean = np.empty(c), np.empty((c, self.ndim))
            _lib.check(self._lib.bobe_gp_predict_grad(self._h, _lib.ptr(x), c, _lib.ptr(mean), None, _lib.ptr(dmean),
                                                      None), "bobe_gp_predict_grad")
            return mean, None, dmean, None
        mean, var = np.empty(c), np.empty(c)
        dmean, dvar = np.empty((c, self.ndim)), np.empty((c, self.ndim))
        _lib.check(self._lib.bobe_gp_predict_grad(self._h, _lib.ptr(x), c, _lib.ptr(mean), _lib.ptr(var),
                                                  _lib.ptr(dmean), _lib.ptr(dvar)), "bobe_gp_predict_grad")
        return mean, var, dmean, dvar

    def hmc_leapfrog(self, U, Pm, inv_mass, eps: float, L: int, temp: float = 1.0):
        """``L`` leapfrog steps of P Hamiltonian-Monte-Carlo chains on the surrogate's mean in ONE GPU launch
        (``bobe_gp_hmc_leapfrog``): U, Pm are (P, d) positions in logit space and momenta (Pm = p0 + eps/2 * grad).
        Returns (U', Pm', logp, grad, mean (physical units), X')."""
        U = _lib.as_f64(np.atleast_2d(U)).copy()
        Pm = _lib.as_f64(np.atleast_2d(Pm)).copy()
        im = _lib.as_f64(inv_mass).reshape(-1)
        P = U.shape[0]
        logp, mean = np.empty(P), np.empty(P)
        grad, X = np.empty_like(U), np.empty_like(U)
        _lib.check(self._lib.bobe_gp_hmc_leapfrog(self._h, P, _lib.ptr(U), _lib.ptr(Pm), _lib.ptr(im), float(eps), int(L),
                                                  float(self.y_std), float(self.y_mean), float(temp), _lib.ptr(logp),
                                                  _lib.ptr(grad), _lib.ptr(mean), _lib.ptr(X)), "bobe_gp_hmc_leapfrog")
        return U, Pm, logp, grad, mean, X

    def hmc_run(self, state, adapt, inv_mass, seed: int, it0: int, niter: int, do_adapt: bool, temp: float = 1.0,
                hist_from: Optional[int] = None, thin: int = 0, debug: bool = False):
        """``niter`` whole HMC trajectories of every chain in ONE GPU launch (``bobe_gp_hmc_run``): ``state`` (P, 3d+2) =
        [u, dlogp/du, x, logp, mean] and ``adapt`` (P, 5) = [eps, mu, hbar, log_eps_bar, m] are updated in place.
        Returns (hist, keep, dbg): u after the iterations >= ``hist_from`` (niter - hist_from, P, d), [x, mean] after
        every ``thin``-th iteration (niter // thin, P, d+1), and the last iteration's draws (P, d+3) - each None
        unless asked for."""
        P, d = state.shape[0], self.ndim
        assert state.shape == (P, 3 * d + 2) and adapt.shape == (P, 5) and state.flags.c_contiguous and adapt.flags.c_contiguous
        im = _lib.as_f64(inv_mass).reshape(-1)
        hist = np.empty((niter - hist_from, P, d)) if hist_from is not None else None
        keep = np.empty((niter // thin, P, d + 1)) if thin else None
        dbg = np.empty((P, d + 3)) if debug else None
        _lib.check(self._lib.bobe_gp_hmc_run(self._h, P, _lib.ptr(state), _lib.ptr(adapt), _lib.ptr(im), int(seed), int(it0),
                                             int(niter), int(bool(do_adapt)), float(self.y_std), float(self.y_mean), float(temp),
                                             int(hist_from or 0), _lib.ptr(hist) if hist is not None else None,
                                             int(thin) if thin else 1, _lib.ptr(keep) if keep is not None else None,
                                             _lib.ptr(dbg) if dbg is not None else None), "bobe_gp_hmc_run")
        return hist, keep, dbg

    def rwalk(self, x, logl, step, lstar: float, walks: int, seed: int, debug: bool = False):
        """Constrained random walks of P walkers on the surrogate's mean in ONE GPU launch (``bobe_gp_rwalk``): the
        replacement search of nested sampling (dynesty's 'rwalk', samplers.py:64, 152).  ``x`` (P, d) start points with
        their physical-unit means ``logl``; ``step`` (d, d) lower-triangular proposal factor; a step is accepted inside the
        unit cube above ``lstar`` (and inside the classifier's region when a gate is set).
        Returns (x', logl', n_accepted, n_inside[, last proposals])."""
        x = _lib.as_f64(np.atleast_2d(x)).copy()
        if x.shape[1] != self.ndim:
            raise ValueError(f"walkers have {x.shape[1]} coordinates, the GP has {self.ndim}")
        lg = _lib.as_f64(logl).reshape(-1).copy()
        st = _lib.as_f64(step).reshape(self.ndim, self.ndim)
        P = x.shape[0]
        nacc, nin = np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32)
        dbg = np.empty_like(x) if debug else None
        _lib.check(self._lib.bobe_gp_rwalk(self._h, P, _lib.ptr(x), _lib.ptr(lg), _lib.ptr(st), float(lstar), int(walks),
                                           int(seed), float(self.y_std), float(self.y_mean), C.c_void_p(nacc.ctypes.data),
                                           C.c_void_p(nin.ctypes.data), _lib.ptr(dbg) if debug else None), "bobe_gp_rwalk")
        return (x, lg, nacc, nin, dbg) if debug else (x, lg, nacc, nin)

    def get_random_point(self, rng=None, nstd=None):
        """BOBE/gp.py:578-585."""
        rng = rng if rng is not None else get_numpy_rng()
        return rng.uniform(0, 1, size=self.train_x.shape[1])

    # ------------------------------------------------------------------ state (gp.py:587-750)
    def state_dict(self, with_factor: bool = True):
        """Same keys as BOBE/gp.py:597-634 (npz-interchangeable).  ``with_factor=False`` (internal, ``copy``) leaves
        the N x N factor on the GPU."""
        return {
            "train_x": np.array(self.train_x),
            "train_y": np.array(self.train_y * self.y_std + self.y_mean),
            "lengthscales": np.array(self.lengthscales),
            "kernel_variance": float(self.kernel_variance),
            "noise": float(self.noise),
            "tausq": float(self.tausq),
            "y_mean": float(self.y_mean),
            "y_std": float(self.y_std),
            "kernel_name": self.kernel_name,
            "lengthscale_prior_spec": self.lengthscale_prior_spec,
            "kernel_variance_prior_spec": self.kernel_variance_prior_spec,
            "fixed_kernel_variance": self.fixed_kernel_variance,
            "optimizer_method": self.optimizer_method,
            "optimizer_options": self.optimizer_options,
            "lengthscale_bounds": self.lengthscale_bounds,
            "kernel_variance_bounds": self.kernel_variance_bounds,
            "tausq_bounds": self.tausq_bounds,
            "cholesky": np.array(self.cholesky) if with_factor else None,
            "alphas": np.array(self.alphas) if with_factor else None,
            "ndim": self.ndim,
            "gp_class": "GP",
        }

    @classmethod
    def from_state_dict(cls, state, device: int = 0, _clone_of=None):
        """BOBE/gp.py:638-677 — L and alpha are restored on the GPU without refactorising (``_clone_of``: take them
        from that GP's device state instead of ``state``)."""
        def plain(v):
            return v.item() if isinstance(v, np.ndarray) and v.shape == () else v
        gp = cls(train_x=state["train_x"], train_y=state["train_y"], noise=plain(state["noise"]),
                 kernel=plain(state["kernel_name"]), optimizer=plain(state["optimizer_method"]),
                 optimizer_options=plain(state["optimizer_options"]), lengthscales=state["lengthscales"],
                 kernel_variance=plain(state["kernel_variance"]),
                 lengthscale_bounds=list(np.asarray(state["lengthscale_bounds"]).tolist()),
                 kernel_variance_bounds=list(np.asarray(state["kernel_variance_bounds"]).tolist()),
                 kernel_variance_prior=plain(state.get("kernel_variance_prior_spec")),
                 lengthscale_prior=plain(state.get("lengthscale_prior_spec")),
                 tausq=plain(state.get("tausq", 1.0)),
                 tausq_bounds=list(np.asarray(state.get("tausq_bounds", [1e-4, 1e4])).tolist()), device=device,
                 _factor=False)
        L, a = state.get("cholesky"), state.get("alphas")
        if _clone_of is not None:
            _lib.check(gp._lib.bobe_gp_clone_state(gp._h, _clone_of._h), "bobe_gp_clone_state")
            gp.not_pd = bool(_clone_of.not_pd)
            # the host copy of the standardisation too, bit for bit (a state_dict round trip re-derives it)
            gp.train_y, gp.y_mean, gp.y_std = np.array(_clone_of.train_y), _clone_of.y_mean, _clone_of.y_std
            gp._pushed_hyper = _clone_of._pushed_hyper       # (the clone carries the source's device hyper-parameters)
            gp._chol_cache, gp._alpha_cache = None, None
        elif L is not None and a is not None and np.all(np.isfinite(np.asarray(L, dtype=np.float64))):
            L = _lib.as_f64(L)
            a = _lib.as_f64(a).reshape(-1)
            gp._push_hyper()
            _lib.check(gp._lib.bobe_gp_set_chol(gp._h, _lib.ptr(L), _lib.ptr(a)), "bobe_gp_set_chol")
            gp._chol_cache, gp._alpha_cache = None, None
        else:
            gp.recompute_cholesky()
        return gp

    @classmethod
    def load(cls, filename, **kwargs):
        """BOBE/gp.py:679-721; further keywords override entries of the stored state, as there.  One keyword is this
        build's own: ``device`` (HIP device of the restored GP, default 0)."""
        device = int(kwargs.pop("device", 0))
        if not filename.endswith(".npz"):
            filename += ".npz"
        data = np.load(filename, allow_pickle=True)
        state = {}
        for key in data.files:
            value = data[key]
            state[key] = value.item() if isinstance(value, np.ndarray) and value.shape == () else value
        state.update(kwargs)
        return cls.from_state_dict(state, device=device)

    def save(self, filename="gp"):
        """BOBE/gp.py:723-737."""
        if not filename.endswith(".npz"):
            filename += ".npz"
        np.savez(filename, **self.state_dict())

    def copy(self):
        """BOBE/gp.py:740-750 — an independent GP with the same data, hyper-parameters and factor.  The reference goes
        through ``state_dict`` / ``from_state_dict``; here the device state is duplicated on the GPU
        (``bobe_gp_clone_state``): no N x N host round trip, no factorisation."""
        state = self.state_dict(with_factor=False)
        new = self.__class__.from_state_dict(state, device=self.device, _clone_of=self)
        return new

    @property
    def npoints(self):
        return self.train_x.shape[0]

    def get_hyperparams(self):
        """BOBE/gp.py:756-762."""
        hp = np.array(self.lengthscales, dtype=np.float64)
        if not self.fixed_kernel_variance:
            hp = np.hstack([hp, self.kernel_variance])
        if self.lengthscale_prior_spec == "SAAS":
            hp = np.hstack([hp, self.tausq])
        return hp

    def hyperparams_dict(self):
        """BOBE/gp.py:764-772."""
        ls_str = {name: f"{float(val):.4f}" for name, val in zip(self.param_names, self.lengthscales)}
        out = {"lengthscales": ls_str, "kernel_variance": f"{float(self.kernel_variance):.4f}"}
        if "tausq" in self.hyperparam_names:
            out["tausq"] = f"{float(self.tausq):.4f}"
        return out
