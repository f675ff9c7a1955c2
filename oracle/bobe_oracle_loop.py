"""CPU restatement of the CALLERS either side of the hot path — TEST INFRASTRUCTURE, never shipped or measured.

Companion of ``bobe_oracle.py`` (same status: a restatement of the reference's algorithm in NumPy/SciPy, every
function citing the reference lines it follows; parity unpinned except where tests/golden/reference_held.json says
otherwise).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

  * kriging-believer batches and the WIPV / WIPStd ``get_next_point``      BOBE/acquisition.py:147-196, 350-412
  * the BO loop's refit policy                                             BOBE/bo.py:620-668
  * dynesty's trapezoid evidence integral and the GP +-sigma logZ bounds   BOBE/samplers.py:27-50, 172-185
  * the classifier gate of GPwithClassifier                                BOBE/clf_gp.py:173-205
  * the SVM-RBF decision function behind it                               BOBE/clf.py:188-213
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np

from . import bobe_oracle as O


# --------------------------------------------------------------------------------------
# acquisition.py:350-412 — WIPV / WIPStd get_next_point on an OracleGP
# --------------------------------------------------------------------------------------
def wip_fun(gp: O.OracleGP, x: np.ndarray, mc_points: np.ndarray, kind: str) -> float:
    """``WIPV.fun`` / ``WIPStd.fun`` (acquisition.py:438-440, 463-465) with the literal ``fantasy_var``."""
    k_train_mc = gp._k12(mc_points)
    var = gp.fantasy_var(np.asarray(x, dtype=np.float64), mc_points, k_train_mc)
    return float(np.mean(var) if kind == "wipv" else np.mean(np.sqrt(var)))


def _fd_value_and_grad(f: Callable[[np.ndarray], float], eps: float = 1e-6):
    """The reference differentiates ``fun`` with jax.grad (acquisition.py:405 -> optim.py:306-309); the oracle has no
    autodiff, so its stand-in is a central difference (clipped to the unit cube like the bounds of the search)."""
    def vg(x):
        x = np.asarray(x, dtype=np.float64)
        g = np.empty_like(x)
        for j in range(x.size):
            hi, lo = x.copy(), x.copy()
            hi[j] = min(1.0, x[j] + eps)
            lo[j] = max(0.0, x[j] - eps)
            g[j] = (f(hi) - f(lo)) / (hi[j] - lo[j])
        return f(x), g
    return vg


def get_next_point(gp: O.OracleGP, kind: str, mc_samples_x: np.ndarray, mc_points_size: int, rng,
                   maxiter: int = 100, refine: bool = True):
    """acquisition.py:350-412: mc_points = rng.choice (acquisition.py:485-489); scores of the candidates (= the
    integration points, acquisition.py:394); argmin (395-398); N > 500 returns it (400-401); otherwise one
    L-BFGS-B run from it in [0,1]^d (403-412, optim.py:249-359 with one restart).
    Returns (x, value, dict(mc_points, sweep_index, sweep_value))."""
    mc_points = O.get_mc_points(mc_samples_x, mc_points_size, rng)
    sw = O.wip_sweep(gp, mc_points, mc_points)
    scores = sw[kind]
    idx = int(np.argmin(scores))
    best_x, best_val = np.array(mc_points[idx]), float(scores[idx])
    info = {"mc_points": mc_points, "sweep_index": idx, "sweep_value": best_val}
    if gp.train_x.shape[0] > 500 or not refine:
        return best_x, best_val, info
    vg = _fd_value_and_grad(lambda x: wip_fun(gp, x, mc_points, kind))
    x, v = O.optimize_scipy(vg, gp.ndim, [0, 1], best_x, maxiter=maxiter, n_restarts=1)
    return np.asarray(x), float(v), info


def get_next_batch(gp: O.OracleGP, kind: str, mc_samples_x: np.ndarray, mc_points_size: int, n_batch: int, rng,
                   maxiter: int = 100, refine: bool = True):
    """Kriging believer (acquisition.py:147-196): a plain GP with the same data and hyper-parameters
    (175-180), ``update(x_next, predict_mean_single(x_next))`` after every member (182, 194)."""
    xs, vals, infos = [], [], []
    x, v, info = get_next_point(gp, kind, mc_samples_x, mc_points_size, rng, maxiter, refine)
    xs.append(x), vals.append(v), infos.append(info)
    if n_batch > 1:
        dummy = O.OracleGP(gp.train_x, gp.train_y * gp.y_std + gp.y_mean, noise=gp.noise, kernel=gp.kernel_name,
                           lengthscales=gp.lengthscales, kernel_variance=gp.kernel_variance)
        dummy.update(x, dummy.predict_mean_single(x))
        for _ in range(1, n_batch):
            x, v, info = get_next_point(dummy, kind, mc_samples_x, mc_points_size, rng, maxiter, refine)
            xs.append(x), vals.append(v), infos.append(info)
            dummy.update(x, dummy.predict_mean_single(x))
    return np.array(xs), np.array(vals), infos


# --------------------------------------------------------------------------------------
# bo.py:620-668 — update_gp: refit policy by training-set size
# --------------------------------------------------------------------------------------
def refit_policy(n_train_before: int, n_since_last_fit: int, n_new: int, fit_n_points: int):
    """bo.py:632-655.  Returns (refit, n_restarts, maxiter, n_since_last_fit_after_counting).  The size classes use
    strict '<' on both sides (bo.py:639, 644): N == 200 and N >= 750 fall into the last branch."""
    n_since = n_since_last_fit + n_new                                   # bo.py:636
    if n_train_before < 200:                                             # bo.py:639-643
        refit_threshold, maxiter, n_restarts = min(2, fit_n_points), 1000, 8
    elif 200 < n_train_before < 750:                                     # bo.py:644-648
        refit_threshold, n_restarts, maxiter = fit_n_points, 4, 500
    else:                                                                # bo.py:649-653
        refit_threshold, n_restarts, maxiter = max(40, fit_n_points), 4, 200
    return n_since >= refit_threshold, n_restarts, maxiter, n_since      # bo.py:655


def update_gp(gp: O.OracleGP, new_x: np.ndarray, new_y: np.ndarray, n_since_last_fit: int, fit_n_points: int, rng):
    """bo.py:620-668 on an OracleGP: count, decide, ``gp.update`` (658), then the multi-restart fit (661-665 ->
    pool.py:268-293).  Returns (n_since_last_fit, refit, n_restarts, maxiter)."""
    refit, n_restarts, maxiter, n_since = refit_policy(gp.train_x.shape[0], n_since_last_fit, np.atleast_2d(new_x).shape[0],
                                                       fit_n_points)
    gp.update(new_x, new_y)
    if refit:
        x0 = O.restart_points(np.log(gp.get_hyperparams()), gp.hyperparam_bounds, n_restarts, rng)
        res = gp.fit(x0=x0, maxiter=maxiter)
        gp.update_hyperparams(res["params"])
        n_since = 0
    return n_since, refit, n_restarts, maxiter


# --------------------------------------------------------------------------------------
# samplers.py:27-50, 172-185 — evidence integral and the GP +-sigma bounds
# --------------------------------------------------------------------------------------
def compute_integrals(logl, logvol, reweight=None, squared=False):
    """samplers.py:27-50: cumulative log-evidence, trapezoid rule in prior volume (logvol_0 = 0)."""
    logl = np.asarray(logl, dtype=np.float64)
    logvol = np.asarray(logvol, dtype=np.float64)
    n = logl.size
    out = np.empty(n)
    acc = -np.inf
    prev_l, prev_v = -1.0e300, 0.0
    for i in range(n):                                    # written as the loop the vectorised original stands for
        dlv = logvol[i] - prev_v                          # log X_i - log X_{i-1}  (<= 0)
        logdvol = logvol[i] - dlv + math.log1p(-math.exp(dlv))            # log(X_{i-1} - X_i)
        if squared:
            logdvol *= 2.0
        logwt = np.logaddexp(logl[i], prev_l) + logdvol + math.log(0.5)   # log( (L_i + L_{i-1}) dX / 2 )
        if reweight is not None:
            logwt += reweight[i]
        acc = np.logaddexp(acc, logwt)
        out[i] = acc
        prev_l, prev_v = logl[i], logvol[i]
    return out


def logz_bounds(logl, logvol, var, mean):
    """samplers.py:172-183: upper / lower logZ from logl +- std, and the variance estimate."""
    logl = np.asarray(logl, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    std = np.sqrt(var)                                                    # samplers.py:173
    upper = compute_integrals(logl + std, logvol)[-1]                     # 174-176
    lower = compute_integrals(logl - std, logvol)[-1]
    var = np.clip(var, 1e-12, 1e12)                                       # 178
    log_var_delta = compute_integrals(2 * logl + np.log(var), logvol, squared=True)[-1]    # 179-180
    log_var_logz = float(np.clip(log_var_delta - 2 * mean, -100, 100))    # 181-182
    var_logz = math.exp(log_var_logz)                                     # 183
    return {"upper": float(upper), "lower": float(lower), "var": var_logz, "std": 2 * math.sqrt(var_logz)}


# --------------------------------------------------------------------------------------
# clf_gp.py:173-205 — the classifier gate
# --------------------------------------------------------------------------------------
def clf_gate(mean, var, clf_probs, probability_threshold: float, minus_inf: float, use_clf: bool = True):
    """``predict_mean_single`` / ``predict_var_single`` / ``predict_single`` of GPwithClassifier: where the
    classifier's probability is below the threshold the mean becomes ``minus_inf`` and the variance the noise floor
    1e-12 (clf_gp.py:179-180, 188-189, 203-205); no classifier, no gate (175-176)."""
    mean = None if mean is None else np.asarray(mean, dtype=np.float64)
    var = None if var is None else np.asarray(var, dtype=np.float64)
    if not use_clf or clf_probs is None:
        return mean, var
    ok = np.asarray(clf_probs) >= probability_threshold
    gm = None if mean is None else np.where(ok, mean, minus_inf)
    gv = None if var is None else np.where(ok, var, O.SAFE_NOISE_FLOOR)
    return gm, gv


def svm_predict(x, support_vectors, dual_coef, intercept: float, gamma: float):
    """clf.py:188-208, line for line, for a batch of points: ``diff = support_vectors - x`` (203), ``norm_sq =
    sum(diff**2, axis=1)`` (204), ``kernel_vals = exp(-gamma * norm_sq)`` (206), ``decision = sum(dual_coef *
    kernel_vals) + intercept`` (208).  Direct differences - NOT the |x|^2 + |sv|^2 - 2 x.sv expansion libsvm uses."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    sv = np.asarray(support_vectors, dtype=np.float64)
    dual = np.asarray(dual_coef, dtype=np.float64).reshape(-1)
    out = np.empty(x.shape[0])
    for c in range(x.shape[0]):
        diff = sv - x[c]
        norm_sq = np.sum(diff ** 2, axis=1)
        kernel_vals = np.exp(-gamma * norm_sq)
        out[c] = np.sum(dual * kernel_vals) + intercept
    return out


def svm_predict_proba(x, support_vectors, dual_coef, intercept: float, gamma: float):
    """clf.py:210-213: 1 where the decision is >= 0, else 0."""
    return np.where(svm_predict(x, support_vectors, dual_coef, intercept, gamma) >= 0, 1.0, 0.0)


def svm_decision_scale(x, support_vectors, dual_coef, gamma: float):
    """sum_i |dual_i| exp(-gamma |sv_i - x|^2): the magnitude the decision's terms cancel from - rounding differences
    between two summation orders are a few ulp of THIS, not of the decision itself (C = 1e7 makes the terms huge)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    sv = np.asarray(support_vectors, dtype=np.float64)
    dual = np.abs(np.asarray(dual_coef, dtype=np.float64).reshape(-1))
    return np.array([np.sum(dual * np.exp(-gamma * np.sum((sv - xc) ** 2, axis=1))) for xc in x])


def clf_labels(train_y_clf: np.ndarray, clf_threshold: float):
    """clf_gp.py:150-153: feasible = within ``clf_threshold`` of the best value seen."""
    y = np.asarray(train_y_clf, dtype=np.float64).reshape(-1)
    return (y >= (np.max(y) - clf_threshold)).astype(int)
