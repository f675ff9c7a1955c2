"""CPU oracle for the BOBE GP hot path (TEST INFRASTRUCTURE — not product code).

This module is a NumPy/SciPy fp64 *restatement* of the reference algorithm in
``BOBE/gp.py``, ``BOBE/acquisition.py``, ``BOBE/optim.py`` and ``BOBE/pool.py``
(Ameek94/BOBE @ 2025-12-26).  Every function cites the reference lines it follows.

PARITY STATUS: **parity unpinned**.  The reference is pure Python on top of
jax / numpyro / tensorflow-probability, none of which is installed in the build
container (ordinary ModuleNotFoundError), and its own tests hold no numeric golden
vectors.  The restatement is therefore pinned only by
  * independent implementations available here (scipy.linalg, scipy.stats,
    scipy.special, torch fp64 autograd, central finite differences),
  * the literal (N+1)-factor ``fantasy_var`` against the rank-1 closed form,
  * the reference tests' invariants re-run on the same data recipes,
  * a second, independently written restatement in plain C (oracle/bobe_oracle_c.c: scalar
    loops, explicit inverse, literal (N+1)-factor fantasy variance) that must agree with
    this one,
see tests/test_oracle.py and tests/golden/make_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker / reported baseline.
The product (``bobe_amd``) never imports it and has no CPU fallback.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular
from scipy.optimize import minimize
from scipy.special import erfcx, ndtr

SAFE_NOISE_FLOOR = 1e-12  # gp.py:16
SQRT2 = math.sqrt(2.0)
SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# kernels  (gp.py:80-168)
# --------------------------------------------------------------------------------------
def dist_sq(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """gp.py:80-96 — sum_j (x_j - y_j)^2 by direct differences (keeps exact zeros)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    out = np.zeros((x.shape[0], y.shape[0]))
    for j in range(x.shape[1]):  # loop over d keeps memory at n1*n2
        diff = x[:, j][:, None] - y[:, j][None, :]
        out += diff * diff
    return out


def kernel_diag(x, kernel_variance, noise, include_noise=True):
    """gp.py:98-122."""
    diag = kernel_variance * np.ones(np.atleast_2d(x).shape[0])
    if include_noise:
        diag = diag + noise
    return diag


def rbf_kernel(xa, xb, lengthscales, kernel_variance, noise, include_noise=True):
    """gp.py:124-154 — sigma^2 exp(-r^2/2) [+ noise*I for the square case]."""
    ls = np.asarray(lengthscales, dtype=np.float64)
    sq = dist_sq(np.asarray(xa) / ls, np.asarray(xb) / ls)
    k = kernel_variance * np.exp(-0.5 * sq)
    if include_noise:
        k = k + noise * np.eye(k.shape[0])
    return k


def matern_kernel(xa, xb, lengthscales, kernel_variance, noise, include_noise=True):
    """gp.py:156-168 — Matern-5/2, r^2 floored at 1e-30 before the sqrt."""
    ls = np.asarray(lengthscales, dtype=np.float64)
    dsq = dist_sq(np.asarray(xa) / ls, np.asarray(xb) / ls)
    d = np.sqrt(np.where(dsq < 1e-30, 1e-30, dsq))
    e = np.exp(-SQRT5 * d)
    poly = 1.0 + d * (SQRT5 + d * 5.0 / 3.0)
    k = kernel_variance * poly * e
    if include_noise:
        k = k + noise * np.eye(k.shape[0])
    return k


def get_kernel(name: str) -> Callable:
    """gp.py:251-252 — anything that is not "rbf" is Matern."""
    return rbf_kernel if name == "rbf" else matern_kernel


# --------------------------------------------------------------------------------------
# Cholesky with XLA-like failure semantics (NaN, no exception)
# --------------------------------------------------------------------------------------
def chol_nan(k: np.ndarray) -> np.ndarray:
    """jnp.linalg.cholesky semantics: lower factor, all-NaN when not positive definite."""
    try:
        if not np.all(np.isfinite(k)):
            raise np.linalg.LinAlgError
        return cholesky(k, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return np.full_like(k, np.nan)


def gp_mll(k: np.ndarray, train_y: np.ndarray, num_points: int) -> float:
    """gp.py:170-178 — returns the (positive) log marginal likelihood data term."""
    L = chol_nan(k)
    if not np.isfinite(L[0, 0]):
        return float("nan")
    y = np.asarray(train_y, dtype=np.float64).reshape(-1, 1)
    alpha = cho_solve((L, True), y, check_finite=False)
    return float(-0.5 * (y.T @ alpha)[0, 0] - np.sum(np.log(np.diag(L))) - 0.5 * num_points * LOG_2PI)


def fast_update_cholesky(L: np.ndarray, k: np.ndarray, k_self: float) -> np.ndarray:
    """gp.py:181-197 — append one row to a Cholesky factor (literal form)."""
    n = L.shape[0]
    v = solve_triangular(L, k, lower=True, check_finite=False)
    with np.errstate(invalid="ignore"):
        diag = np.sqrt(k_self - np.dot(v, v))
    new_L = np.zeros((n + 1, n + 1))
    new_L[:n, :n] = L
    new_L[n, :n] = v
    new_L[n, n] = diag
    return new_L


# --------------------------------------------------------------------------------------
# priors  (numpyro 0.15.3 log-densities restated; gp.py:56-78, 309-366)
# --------------------------------------------------------------------------------------
def _logpdf_uniform(x, low, high):
    # numpyro Uniform.log_prob: -log(high-low) broadcast to x (no support masking by default)
    return -np.log(high - low) * np.ones_like(np.asarray(x, dtype=np.float64))


def _logpdf_normal(x, loc, scale):
    x = np.asarray(x, dtype=np.float64)
    return -0.5 * ((x - loc) / scale) ** 2 - np.log(scale) - 0.5 * LOG_2PI


def _logpdf_lognormal(x, loc, scale):
    x = np.asarray(x, dtype=np.float64)
    return _logpdf_normal(np.log(x), loc, scale) - np.log(x)


def _logpdf_halfcauchy(x, scale):
    x = np.asarray(x, dtype=np.float64)
    return np.log(2.0) - np.log(np.pi) - np.log(scale) - np.log1p((x / scale) ** 2)


def _logpdf_halfnormal(x, scale):
    x = np.asarray(x, dtype=np.float64)
    return _logpdf_normal(x, 0.0, scale) + np.log(2.0)


def _logpdf_gamma(x, concentration, rate=1.0):
    from scipy.special import gammaln
    x = np.asarray(x, dtype=np.float64)
    return concentration * np.log(rate) + (concentration - 1) * np.log(x) - rate * x - gammaln(concentration)


_DISTS = {
    "Uniform": lambda x, low=0.0, high=1.0: _logpdf_uniform(x, low, high),
    "Normal": lambda x, loc=0.0, scale=1.0: _logpdf_normal(x, loc, scale),
    "LogNormal": lambda x, loc=0.0, scale=1.0: _logpdf_lognormal(x, loc, scale),
    "HalfCauchy": lambda x, scale=1.0: _logpdf_halfcauchy(x, scale),
    "HalfNormal": lambda x, scale=1.0: _logpdf_halfnormal(x, scale),
    "Gamma": lambda x, concentration=1.0, rate=1.0: _logpdf_gamma(x, concentration, rate),
}


def make_distribution(spec: dict) -> Callable:
    """gp.py:27-54 — dict spec -> log_prob callable."""
    name = spec["name"]
    if name not in _DISTS:
        raise ValueError(f"Distribution {name} not found")
    kwargs = {k: v for k, v in spec.items() if k != "name"}
    f = _DISTS[name]
    return lambda x: f(x, **kwargs)


def saas_prior_logprob(lengthscales, kernel_variance, tausq) -> float:
    """gp.py:56-78."""
    lp = _logpdf_lognormal(kernel_variance, 0.0, 1.0)
    lp = lp + _logpdf_halfcauchy(tausq, 0.1)
    inv_ls_sq = 1.0 / (tausq * np.asarray(lengthscales) ** 2)
    lp = lp + np.sum(_logpdf_halfcauchy(inv_ls_sq, 1.0))
    return float(lp)


# --------------------------------------------------------------------------------------
# analytic gradient of the MLL data term (reference: jax.value_and_grad, optim.py:306-309)
# --------------------------------------------------------------------------------------
def mll_value_and_grad(kernel_name, X, y, lengthscales, kernel_variance, noise):
    """Data-term MLL (gp.py:170-178) and d/d(log ls_j), d/d(log kvar).

    With W = alpha alpha^T - K^-1 and Kt = K without the noise term,
      dMLL/dlog ls_j = 1/2 sum_ab W_ab dKt_ab/dlog ls_j ,  dMLL/dlog kvar = 1/2 sum_ab W_ab Kt_ab.
    RBF:    dKt/dlog ls_j = Kt * D_j,                D_j = ((x_aj-x_bj)/ls_j)^2
    Matern: dKt/dlog ls_j = kvar*(5/3)(1+sqrt5 r)exp(-sqrt5 r) * D_j   (zero where r^2 < 1e-30)
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    ls = np.asarray(lengthscales, dtype=np.float64)
    n, d = X.shape
    kern = get_kernel(kernel_name)
    Kt = kern(X, X, ls, kernel_variance, noise, include_noise=False)
    K = Kt + noise * np.eye(n)
    L = chol_nan(K)
    if not np.isfinite(L[0, 0]):
        return float("nan"), np.full(d + 1, np.nan)
    alpha = cho_solve((L, True), y, check_finite=False)
    mll = float(-0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * n * LOG_2PI)
    Kinv = cho_solve((L, True), np.eye(n), check_finite=False)
    W = np.outer(alpha, alpha) - Kinv
    grad = np.empty(d + 1)
    Xs = X / ls
    if kernel_name == "rbf":
        WK = W * Kt
        for j in range(d):
            diff = Xs[:, j][:, None] - Xs[:, j][None, :]
            grad[j] = 0.5 * np.sum(WK * diff * diff)
        grad[d] = 0.5 * np.sum(WK)
    else:
        dsq = dist_sq(Xs, Xs)
        r = np.sqrt(np.where(dsq < 1e-30, 1e-30, dsq))
        G = kernel_variance * (5.0 / 3.0) * (1.0 + SQRT5 * r) * np.exp(-SQRT5 * r)
        G = np.where(dsq < 1e-30, 0.0, G)
        WG = W * G
        for j in range(d):
            diff = Xs[:, j][:, None] - Xs[:, j][None, :]
            grad[j] = 0.5 * np.sum(WG * diff * diff)
        grad[d] = 0.5 * np.sum(W * Kt)
    return mll, grad


# --------------------------------------------------------------------------------------
# optimiser driver  (optim.py:249-359) and restart recipe (pool.py:268-286)
# --------------------------------------------------------------------------------------
def setup_bounds(bounds, num_params):
    """optim.py:42-68."""
    if bounds is None:
        return None
    bounds = np.array(bounds, dtype=np.float64)
    if bounds.shape == (2,):
        bounds = np.tile(bounds.reshape(1, 2), (num_params, 1)).T
    elif bounds.shape != (2, num_params):
        raise ValueError(f"Bounds shape {bounds.shape} incompatible with {num_params} parameters")
    return bounds


def optimize_scipy(value_and_grad, num_params, bounds, x0, maxiter=200, n_restarts=4,
                   optimizer_options: Optional[dict] = None):
    """optim.py:249-359 — restart loop, start-point screening, acceptance rule.

    ``value_and_grad(x) -> (f, g)`` replaces the jitted ``jax.value_and_grad(fun)``
    closure of optim.py:306-309.
    """
    options = dict(optimizer_options or {})
    options.update({"maxiter": maxiter})                      # optim.py:292
    method = options.pop("method", "L-BFGS-B")                # optim.py:294
    bounds_arr = setup_bounds(bounds, num_params)
    scipy_bounds = None if bounds_arr is None else [
        (float(bounds_arr[0, i]), float(bounds_arr[1, i])) for i in range(num_params)]
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    if x0.shape[0] < n_restarts:
        raise ValueError(f"x0 provided with {x0.shape[0]} restarts but n_restarts={n_restarts}")
    x0 = x0[:n_restarts]
    best_f, best_x = np.inf, None
    for x_init in x0:                                         # optim.py:325-333
        val, _ = value_and_grad(x_init)
        if np.isfinite(val) and val < best_f:
            best_f, best_x = float(val), np.array(x_init)
    for x_init in x0:                                         # optim.py:335-354
        try:
            res = minimize(value_and_grad, x_init, method=method, jac=True,
                           bounds=scipy_bounds, options=options)
        except Exception:
            continue
        ok = res.success or "ITERATIONS REACHED LIMIT" in str(res.message).upper()
        if ok and np.isfinite(res.fun) and res.fun < best_f:
            best_f, best_x = float(res.fun), np.array(res.x)
    return best_x, float(best_f)


def restart_points(log_hyper, log_bounds, n_restarts, rng):
    """pool.py:277-286 — row 0 = log(current hp), rows 1.. uniform in the log-bounds."""
    log_hyper = np.asarray(log_hyper, dtype=np.float64)
    if n_restarts > 1:
        x0_random = rng.uniform(log_bounds[0], log_bounds[1], size=(n_restarts - 1, log_hyper.shape[0]))
        return np.vstack([log_hyper, x0_random])
    return np.atleast_2d(log_hyper)


# --------------------------------------------------------------------------------------
# EI / LogEI scorers (acquisition.py:21-75, 226-253, 318-330)
# --------------------------------------------------------------------------------------
def _norm_pdf(u):
    return np.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)


def ei_helper(u):
    """acquisition.py:29-31."""
    return _norm_pdf(u) + u * ndtr(u)


def log1mexp(x):
    """tfp.math.log1mexp: log(1 - exp(-|x|)), switch at log 2."""
    x = np.abs(np.asarray(x, dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(x < math.log(2.0), np.log(-np.expm1(-x)), np.log1p(-np.exp(-x)))


def log_ei_helper(u):
    """acquisition.py:44-75 (float64 branch: neg_inv_sqrt_eps = -1e6)."""
    u = np.asarray(u, dtype=np.float64)
    bound = -1.0
    neg_inv_sqrt_eps = -1e6
    u_upper = np.where(u < bound, bound, u)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_ei_upper = np.log(ei_helper(u_upper))
        u_lower = np.where(u > bound, bound, u)
        u_eps = np.where(u_lower < neg_inv_sqrt_eps, neg_inv_sqrt_eps, u_lower)
        # acquisition.py:33-42
        w = np.log(np.abs(u_eps) * erfcx(-u_eps / SQRT2)) + 0.5 * math.log(math.pi / 2.0)
        log_phi_u = -0.5 * (u * u + LOG_2PI)
        second = np.where(u > neg_inv_sqrt_eps, log1mexp(w), -2.0 * np.log(np.abs(u_lower)))
    return np.where(u > bound, log_ei_upper, log_phi_u + second)


def ei_score(mu, var, best_y, zeta=0.0):
    """acquisition.py:226-253 — returns +EI (the reference minimises -EI)."""
    var = np.clip(var, 1e-20, None)
    sigma = np.sqrt(var)
    u = (mu - zeta - best_y) / sigma
    return ei_helper(u) * sigma


def log_ei_score(mu, var, best_y, zeta=0.0):
    """acquisition.py:318-330 — returns +log EI."""
    var = np.clip(var, 1e-18, None)
    sigma = np.sqrt(var)
    u = (mu - zeta - best_y) / sigma
    return log_ei_helper(u) + np.log(sigma)


# --------------------------------------------------------------------------------------
# the GP object (gp.py:199-772), CPU restatement
# --------------------------------------------------------------------------------------
class OracleGP:
    """CPU twin of ``BOBE.gp.GP`` with the same attribute / method names."""

    def __init__(self, train_x, train_y, noise=1e-8, kernel="rbf", optimizer="scipy", optimizer_options=None,
                 kernel_variance_bounds=(1e-4, 1e8), lengthscale_bounds=(0.01, 5), lengthscales=None,
                 kernel_variance=None, kernel_variance_prior=None, lengthscale_prior=None, tausq=None,
                 tausq_bounds=(1e-4, 1e4), param_names=None):
        self._setup_training_data(train_x, train_y)                       # gp.py:247
        self.param_names = param_names if param_names is not None else [f"x_{i}" for i in range(self.ndim)]
        self.kernel_name = kernel if kernel == "rbf" else "matern"        # gp.py:251
        self.kernel = get_kernel(self.kernel_name)
        self.lengthscales = (np.asarray(lengthscales, dtype=np.float64) if lengthscales is not None
                             else np.ones(self.ndim))
        self.kernel_variance = float(kernel_variance) if kernel_variance is not None else 1.0
        self.noise = float(noise)
        self.optimizer_method = optimizer
        self.optimizer_options = dict(optimizer_options or {})
        self.lengthscale_bounds = list(lengthscale_bounds)
        self.kernel_variance_bounds = list(kernel_variance_bounds)
        self.tausq = float(tausq) if tausq is not None else 1.0
        self.tausq_bounds = list(tausq_bounds)
        self._setup_priors(kernel_variance_prior, lengthscale_prior)
        self._setup_optimization_parameters()
        self.recompute_cholesky()                                         # gp.py:257-260

    # -- data -------------------------------------------------------------------------
    def _setup_training_data(self, train_x, train_y):
        """gp.py:283-307 — population std; std==0 -> 1."""
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
            self.y_std = 1.0
        self.train_x = np.array(train_x)
        self.train_y = (train_y - self.y_mean) / self.y_std

    # -- priors -----------------------------------------------------------------------
    def _setup_priors(self, kernel_variance_prior, lengthscale_prior):
        """gp.py:309-337."""
        self.kernel_variance_prior_spec = kernel_variance_prior
        if self.kernel_variance_prior_spec is None:
            self.kernel_variance_prior_spec = {"name": "Uniform", "low": self.kernel_variance_bounds[0],
                                               "high": self.kernel_variance_bounds[1]}
        self.fixed_kernel_variance = (self.kernel_variance_prior_spec == "fixed")
        self._kvar_logprob = ((lambda x: 0.0) if self.fixed_kernel_variance
                              else make_distribution(self.kernel_variance_prior_spec))
        self.lengthscale_prior_spec = lengthscale_prior
        if self.lengthscale_prior_spec is None:
            self.lengthscale_prior_spec = {"name": "Uniform", "low": self.lengthscale_bounds[0],
                                           "high": self.lengthscale_bounds[1]}
        if self.lengthscale_prior_spec == "DSLP":
            loc, scale = SQRT2 + 0.5 * math.log(self.ndim), SQRT3          # gp.py:330
            self._ls_logprob = lambda x: _logpdf_lognormal(x, loc, scale)
        elif self.lengthscale_prior_spec == "SAAS":
            self._ls_logprob = None
        else:
            self._ls_logprob = make_distribution(self.lengthscale_prior_spec)

    def prior_logprob(self, lengthscales, kernel_variance, tausq) -> float:
        """gp.py:357-366."""
        if self.lengthscale_prior_spec == "SAAS":
            return saas_prior_logprob(lengthscales, kernel_variance, tausq)
        lp = float(np.sum(self._kvar_logprob(kernel_variance)))
        lp += float(np.sum(self._ls_logprob(lengthscales)))
        return lp

    def _setup_optimization_parameters(self):
        """gp.py:339-355."""
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

    def _parse_hyperparams(self, log_params):
        """gp.py:368-383."""
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

    # -- objective --------------------------------------------------------------------
    def neg_mll(self, log_params) -> float:
        """gp.py:385-398."""
        ls, kvar, tausq = self._parse_hyperparams(log_params)
        K = self.kernel(self.train_x, self.train_x, ls, kvar, self.noise, include_noise=True)
        mll = gp_mll(K, self.train_y, self.train_y.shape[0])
        mll += self.prior_logprob(ls, kvar, tausq)
        return -mll

    def _prior_grad_fd(self, log_params, eps=1e-6):
        """d prior / d theta by central differences on the (cheap, O(d)) prior only."""
        g = np.zeros(len(log_params))
        for i in range(len(log_params)):
            tp = np.array(log_params, dtype=np.float64)
            tm = tp.copy()
            tp[i] += eps
            tm[i] -= eps
            g[i] = (self.prior_logprob(*self._parse_hyperparams(tp)) -
                    self.prior_logprob(*self._parse_hyperparams(tm))) / (2 * eps)
        return g

    def neg_mll_value_and_grad(self, log_params):
        """value + gradient of gp.py:385-398 wrt theta = log hp (optim.py:306-309's closure)."""
        log_params = np.asarray(log_params, dtype=np.float64)
        ls, kvar, tausq = self._parse_hyperparams(log_params)
        mll, g_data = mll_value_and_grad(self.kernel_name, self.train_x, self.train_y, ls, kvar, self.noise)
        grad = np.zeros(len(log_params))
        grad[:self.ndim] = g_data[:self.ndim]
        if not self.fixed_kernel_variance:
            grad[self.ndim] = g_data[self.ndim]
        val = mll + self.prior_logprob(ls, kvar, tausq)
        grad = grad + self._prior_grad_fd(log_params)
        return -val, -grad

    def fit(self, x0=None, maxiter=500):
        """gp.py:400-437."""
        if x0 is None:
            x0 = np.log(self.get_hyperparams())[None, :]
        x0 = np.atleast_2d(x0)
        best, loss = optimize_scipy(self.neg_mll_value_and_grad, self.num_hyperparams, self.hyperparam_bounds,
                                    x0, maxiter=maxiter, n_restarts=x0.shape[0],
                                    optimizer_options=dict(self.optimizer_options))
        return {"mll": -loss, "params": best}

    def update_hyperparams(self, hyperparams):
        """gp.py:439-448."""
        ls, kvar, tausq = self._parse_hyperparams(hyperparams)
        self.lengthscales = np.array(ls)
        if not self.fixed_kernel_variance:
            self.kernel_variance = kvar
        self.tausq = tausq
        self.recompute_cholesky()

    def recompute_cholesky(self):
        """gp.py:544-550."""
        K = self.kernel(self.train_x, self.train_x, self.lengthscales, self.kernel_variance, self.noise,
                        include_noise=True)
        self.cholesky = chol_nan(K)
        if np.isfinite(self.cholesky[0, 0]):
            self.alphas = cho_solve((self.cholesky, True), self.train_y, check_finite=False)
        else:
            self.alphas = np.full_like(self.train_y, np.nan)

    # -- prediction -------------------------------------------------------------------
    def _k12(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        return self.kernel(self.train_x, x, self.lengthscales, self.kernel_variance, self.noise, include_noise=False)

    def predict_mean_batched(self, x):
        """gp.py:450-457, 468-470."""
        return (self._k12(x).T @ self.alphas).reshape(-1) * self.y_std + self.y_mean

    def predict_mean_single(self, x):
        return float(self.predict_mean_batched(x)[0])

    def predict_var_batched(self, x):
        """gp.py:459-466, 472-474 — clip keeps NaN."""
        vv = solve_triangular(self.cholesky, self._k12(x), lower=True, check_finite=False)
        var = (self.kernel_variance + self.noise) - np.sum(vv * vv, axis=0)
        var = np.clip(var, SAFE_NOISE_FLOOR, None)
        return self.y_std ** 2 * var

    def predict_var_single(self, x):
        return float(self.predict_var_batched(x)[0])

    def predict_batched(self, x):
        """gp.py:476-493 — standardised (mu, var); NaN and < floor -> floor."""
        k12 = self._k12(x)
        mean = (k12.T @ self.alphas).reshape(-1)
        vv = solve_triangular(self.cholesky, k12, lower=True, check_finite=False)
        var = (self.kernel_variance + self.noise) - np.sum(vv * vv, axis=0)
        var = np.where(np.isnan(var), SAFE_NOISE_FLOOR, var)
        var = np.where(var < SAFE_NOISE_FLOOR, SAFE_NOISE_FLOOR, var)
        return mean, var

    def predict_single(self, x):
        m, v = self.predict_batched(x)
        return float(m[0]), float(v[0])

    # -- update -----------------------------------------------------------------------
    def update(self, new_x, new_y):
        """gp.py:495-541 — duplicate filter, re-standardise, full refactor."""
        new_x = np.atleast_2d(np.asarray(new_x, dtype=np.float64))
        new_y = np.atleast_2d(np.asarray(new_y, dtype=np.float64))
        pts, vals = [], []
        for i in range(new_x.shape[0]):
            if np.any(np.all(np.isclose(self.train_x, new_x[i], atol=1e-6, rtol=1e-4), axis=1)):
                continue
            pts.append(new_x[i])
            vals.append(new_y[i])
        if pts:
            self.train_x = np.vstack([self.train_x, np.array(pts)])
            y_orig = np.vstack([self.train_y * self.y_std + self.y_mean, np.array(vals).reshape(-1, 1)])
            self.y_mean = float(np.mean(y_orig))
            self.y_std = float(np.std(y_orig))
            if self.y_std == 0:
                self.y_std = 1.0
            self.train_y = (y_orig - self.y_mean) / self.y_std
            self.recompute_cholesky()

    # -- fantasy variance -------------------------------------------------------------
    def fantasy_var(self, new_x, mc_points, k_train_mc):
        """gp.py:552-576 — LITERAL (N+1)-factor form."""
        new_x = np.atleast_2d(np.asarray(new_x, dtype=np.float64))
        k = self._k12(new_x).flatten()
        k_self = self.kernel_variance + self.noise
        k11_cho = fast_update_cholesky(self.cholesky, k, k_self)
        k_new_mc = self.kernel(new_x, mc_points, self.lengthscales, self.kernel_variance, self.noise,
                               include_noise=False)
        k12 = np.vstack([k_train_mc, k_new_mc])
        k22 = kernel_diag(mc_points, self.kernel_variance, self.noise, include_noise=True)
        with np.errstate(all="ignore"):
            vv = solve_triangular(k11_cho, k12, lower=True, check_finite=False)
            var = k22 - np.sum(vv * vv, axis=0)
        var = np.where(np.isnan(var), SAFE_NOISE_FLOOR, var)
        var = np.where(var < SAFE_NOISE_FLOOR, SAFE_NOISE_FLOOR, var)
        return var * self.y_std ** 2

    def get_random_point(self, rng=None, nstd=None):
        """gp.py:578-585."""
        rng = rng if rng is not None else np.random.default_rng()
        return rng.uniform(0, 1, size=self.train_x.shape[1])

    @property
    def npoints(self):
        return self.train_x.shape[0]

    def get_hyperparams(self):
        """gp.py:756-762."""
        hp = np.array(self.lengthscales, dtype=np.float64)
        if not self.fixed_kernel_variance:
            hp = np.hstack([hp, self.kernel_variance])
        if self.lengthscale_prior_spec == "SAAS":
            hp = np.hstack([hp, self.tausq])
        return hp


# --------------------------------------------------------------------------------------
# WIPV / WIPStd sweep (acquisition.py:350-412, 438-465) — rank-1 closed form
# --------------------------------------------------------------------------------------
def wip_sweep(gp: OracleGP, cand: np.ndarray, mc_points: np.ndarray, chunk: int = 4096):
    """Scores of every candidate against the integration points, rank-1 form.

    Algebraically identical to mapping ``WIPV.fun`` / ``WIPStd.fun``
    (acquisition.py:438-440, 463-465) over the candidates with the literal
    ``fantasy_var`` (gp.py:552-576):
        V_Z = L^-1 K(X,Z); v_c = L^-1 K(X,x_c); s_c = (kvar+noise) - |v_c|^2
        cross(c,z) = K(x_c,z) - v_c . V_Z[:,z]
        var+(z|c)  = (kvar+noise) - |V_Z[:,z]|^2 - cross^2/s_c  -> NaN/<1e-12 -> 1e-12 -> * y_std^2
    s_c < 0 makes the reference's sqrt NaN, which floors every var+(.|c).
    Returns dict(mean, var (standardised, floored as predict_single), wipv, wipstd, argmin_v, argmin_s).
    """
    cand = np.atleast_2d(np.asarray(cand, dtype=np.float64))
    Z = np.atleast_2d(np.asarray(mc_points, dtype=np.float64))
    L = gp.cholesky
    kz = gp._k12(Z)
    VZ = solve_triangular(L, kz, lower=True, check_finite=False)
    kself = gp.kernel_variance + gp.noise
    base = kself - np.sum(VZ * VZ, axis=0)
    C = cand.shape[0]
    out = {k: np.empty(C) for k in ("mean", "var", "wipv", "wipstd")}
    alpha = gp.alphas.reshape(-1)
    for s in range(0, C, chunk):
        xc = cand[s:s + chunk]
        kc = gp._k12(xc)
        vc = solve_triangular(L, kc, lower=True, check_finite=False)
        q = np.sum(vc * vc, axis=0)
        sc = kself - q
        kcz = gp.kernel(xc, Z, gp.lengthscales, gp.kernel_variance, gp.noise, include_noise=False)
        cross = kcz - vc.T @ VZ
        with np.errstate(all="ignore"):
            var = base[None, :] - cross * cross / sc[:, None]
        var = np.where(sc[:, None] >= 0, var, np.nan)          # sqrt(negative) -> NaN (gp.py:187)
        var = np.where(np.isnan(var), SAFE_NOISE_FLOOR, var)
        var = np.where(var < SAFE_NOISE_FLOOR, SAFE_NOISE_FLOOR, var)
        var = var * gp.y_std ** 2
        out["wipv"][s:s + chunk] = np.mean(var, axis=1)
        out["wipstd"][s:s + chunk] = np.mean(np.sqrt(var), axis=1)
        out["mean"][s:s + chunk] = kc.T @ alpha
        pv = np.where(np.isnan(sc), SAFE_NOISE_FLOOR, sc)
        out["var"][s:s + chunk] = np.where(pv < SAFE_NOISE_FLOOR, SAFE_NOISE_FLOOR, pv)
    out["argmin_v"] = int(np.argmin(out["wipv"]))
    out["argmin_s"] = int(np.argmin(out["wipstd"]))
    return out


def wip_sweep_literal(gp: OracleGP, cand: np.ndarray, mc_points: np.ndarray):
    """acquisition.py:388-398 with the literal fantasy_var — O(C N^2 M); small sizes only."""
    Z = np.atleast_2d(np.asarray(mc_points, dtype=np.float64))
    k_train_mc = gp._k12(Z)
    wipv = np.array([np.mean(gp.fantasy_var(x, Z, k_train_mc)) for x in np.atleast_2d(cand)])
    wipstd = np.array([np.mean(np.sqrt(gp.fantasy_var(x, Z, k_train_mc))) for x in np.atleast_2d(cand)])
    return wipv, wipstd


def get_mc_points(mc_samples_x: np.ndarray, mc_points_size: int, rng) -> np.ndarray:
    """acquisition.py:485-489."""
    mc_size = max(mc_samples_x.shape[0], mc_points_size)
    idxs = rng.choice(mc_size, size=mc_points_size, replace=False)
    return mc_samples_x[idxs]


# --------------------------------------------------------------------------------------
# the benchmark "cycle" on the CPU — bench.py's cpu_baseline leg
# --------------------------------------------------------------------------------------
def cycle_value_and_grad(X, y, ls, kvar, noise, kernel="rbf"):
    """One value+grad evaluation of the data-term MLL the way a NumPy/LAPACK port does it: dpotrf + dpotrs + dpotri
    (N^3 flops) and d+1 fused N^2 reductions.  Same formulas as ``mll_value_and_grad`` (gp.py:124-178 under
    optim.py:306-309), with K^-1 from dpotri instead of N solves — the form that is affordable at N of several thousand."""
    from scipy.linalg import lapack
    n, d = X.shape
    Xs = X / ls
    dsq = dist_sq(Xs, Xs)
    if kernel == "rbf":
        Kt = kvar * np.exp(-0.5 * dsq)
        G = Kt                                   # dKt/dlog ls_j = Kt * D_j
    else:                                        # Matern-5/2 (gp.py:156-168), r^2 floored at 1e-30 before the sqrt
        r = np.sqrt(np.where(dsq < 1e-30, 1e-30, dsq))
        e = np.exp(-SQRT5 * r)
        Kt = kvar * (1.0 + r * (SQRT5 + r * 5.0 / 3.0)) * e
        G = np.where(dsq < 1e-30, 0.0, kvar * (5.0 / 3.0) * (1.0 + SQRT5 * r) * e)
    del dsq
    K = Kt + noise * np.eye(n)
    L, info = lapack.dpotrf(K, lower=1, clean=1, overwrite_a=0)
    if info != 0:
        return float("nan"), np.full(d + 1, np.nan)
    alpha, _ = lapack.dpotrs(L, y, lower=1)
    mll = float(-0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * n * LOG_2PI)
    Kinv, _ = lapack.dpotri(L, lower=1)
    Kinv = np.tril(Kinv) + np.tril(Kinv, -1).T
    W = np.outer(alpha, alpha) - Kinv
    WG = W * G
    g = np.empty(d + 1)
    for j in range(d):
        diff = Xs[:, j][:, None] - Xs[:, j][None, :]
        g[j] = 0.5 * np.sum(WG * diff * diff)
    g[d] = 0.5 * np.sum(W * Kt)
    return mll, g
