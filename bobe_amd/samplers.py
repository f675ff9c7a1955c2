"""Evidence (logZ) and MC-point consumers of the GPU GP — counterpart of ``BOBE/samplers.py``.

The reference hands the jitted ``gp.predict_mean_single`` to dynesty, one point per call
(samplers.py:112-115, 157-160) — the worst pattern for a GPU.  Here a static nested sampler runs on the host
and asks the GP for *batches*: replacement candidates are drawn uniformly inside an enlarged bounding
ellipsoid of the live points (clipped to the unit cube), a few thousand at a time, and scored with ONE
``bobe_gp_predict`` call.  Everything downstream follows the reference:
  * ``compute_integrals``  — the dynesty trapezoid evidence integral (samplers.py:27-50);
  * logZ bounds from the GP standard deviation at the samples, logl +- std  (samplers.py:172-176), and the
    variance estimate (samplers.py:178-183);  the BO loop's convergence test is (upper-lower)/2 < threshold
    (bo.py:886-891);
  * result dictionaries with the reference's keys (samplers.py:184-194).
dynesty and NumPyro themselves are not re-implemented (no multi-ellipsoid decomposition, no NUTS).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

from .utils import get_logger, get_numpy_rng

log = get_logger("sampler")


def compute_integrals(logl, logvol, reweight=None, squared=False):
    """samplers.py:27-50 (dynesty utility): cumulative log-evidence by the trapezoid rule in prior volume."""
    logl = np.asarray(logl, dtype=np.float64)
    logvol = np.asarray(logvol, dtype=np.float64)
    loglstar_pad = np.concatenate([[-1.0e300], logl])
    dlogvol = np.diff(logvol, prepend=0)
    logdvol = logvol - dlogvol + np.log1p(-np.exp(dlogvol))
    if squared:
        logdvol = 2 * logdvol
    logdvol2 = logdvol + math.log(0.5)
    saved_logwt = np.logaddexp(loglstar_pad[1:], loglstar_pad[:-1]) + logdvol2
    if reweight is not None:
        saved_logwt = saved_logwt + reweight
    return np.logaddexp.accumulate(saved_logwt)


def renormalise_log_weights(logw):
    """utils/core.py counterpart: exp(logw - logsumexp(logw))."""
    logw = np.asarray(logw, dtype=np.float64)
    m = np.max(logw)
    w = np.exp(logw - m)
    return w / np.sum(w)


def resample_equal(samples, logl, weights, rng=None):
    """Systematic resampling to equal weights (dynesty.utils.resample_equal, used at samplers.py:187-188)."""
    rng = rng if rng is not None else get_numpy_rng()
    n = len(weights)
    positions = (rng.random() + np.arange(n)) / n
    cum = np.cumsum(weights)
    cum[-1] = 1.0
    idx = np.searchsorted(cum, positions)
    return samples[idx], logl[idx]


def _bounding_ellipsoid(live: np.ndarray, enlarge: float):
    """Mean, Cholesky factor of the scaled covariance and its volume factor: {x : |A^-1 (x-mu)| <= 1}."""
    mu = live.mean(axis=0)
    d = live.shape[1]
    cov = np.cov(live, rowvar=False).reshape(d, d) + 1e-12 * np.eye(d)
    L = np.linalg.cholesky(cov)
    z = np.linalg.solve(L, (live - mu).T)
    r = math.sqrt(float(np.max(np.sum(z * z, axis=0)))) * enlarge
    return mu, L * r


def _draw_in_ellipsoid(mu, A, n, rng):
    d = mu.shape[0]
    g = rng.standard_normal((n, d))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    u = rng.random(n) ** (1.0 / d)
    x = mu + (g * u[:, None]) @ A.T
    return x[np.all((x >= 0.0) & (x <= 1.0), axis=1)]


def nested_sampling(gp, ndim: Optional[int] = None, mode: str = "convergence", dlogz: float = 0.01,
                    maxcall: int = int(5e6), equal_weights: bool = False, rng=None, batch: int = 8192,
                    enlarge: float = 1.25, nlive: Optional[int] = None) -> Tuple[Dict, Dict, bool]:
    """Static nested sampling of exp(GP mean) over the unit cube -> (samples_dict, logz_dict, success).

    Settings follow ``nested_sampling_Dy`` (samplers.py:119-126): mode 'acq' uses nlive = max(100, min(500, 20 d))
    and dlogz = 0.1 with equal-weight samples; otherwise nlive = max(500, 40 d)."""
    rng = rng if rng is not None else get_numpy_rng()
    ndim = ndim if ndim is not None else gp.ndim
    if mode == "acq":
        nlive = nlive or max(100, min(500, 20 * ndim))
        dlogz, equal_weights = 0.1, True
    else:
        nlive = nlive or max(500, 40 * ndim)

    def loglike(x):
        return np.asarray(gp.predict_mean_batched(x), dtype=np.float64)

    live = rng.uniform(size=(nlive, ndim))
    live_logl = loglike(live)
    ncall = nlive
    dead_x, dead_logl = [], []
    logz = -np.inf
    pool_x = np.empty((0, ndim))
    pool_l = np.empty(0)
    pool_pos = 0
    since_update = 0
    it = 0
    update_every = max(1, nlive // 5)
    while True:
        worst = int(np.argmin(live_logl))
        lstar = float(live_logl[worst])
        logdx = -it / nlive + math.log1p(-math.exp(-1.0 / nlive))      # log(X_{i-1} - X_i)
        logz_new = np.logaddexp(logz, lstar + logdx)
        dead_x.append(live[worst].copy())
        dead_logl.append(lstar)
        logz = logz_new
        it += 1
        # remaining evidence bound (dynesty's stopping rule): dlogz = log(z + Lmax X) - log z
        lmax = float(np.max(live_logl))
        if np.logaddexp(logz, lmax - it / nlive) - logz < dlogz or ncall >= maxcall:
            break
        # replacement with L > L*: pop pre-scored proposals, refill the pool in GPU batches
        found = False
        tries = 0
        while not found:
            while pool_pos < len(pool_l):
                if pool_l[pool_pos] > lstar:
                    live[worst] = pool_x[pool_pos]
                    live_logl[worst] = pool_l[pool_pos]
                    pool_pos += 1
                    found = True
                    break
                pool_pos += 1
            if found:
                break
            if since_update >= update_every or tries > 0 or len(pool_l) == 0:
                mask = np.ones(nlive, dtype=bool)
                mask[worst] = False
                mu, A = _bounding_ellipsoid(live[mask], enlarge)
                since_update = 0
            x = _draw_in_ellipsoid(mu, A, batch, rng)
            if len(x) == 0:
                x = rng.uniform(size=(batch, ndim))
            pool_x, pool_l, pool_pos = x, loglike(x), 0
            ncall += len(x)
            tries += 1
            if tries > 200:                                   # plateau / degenerate surrogate: give up cleanly
                log.warning("nested sampling: no acceptable replacement found; stopping early")
                found = True
                ncall = maxcall
        since_update += 1
    # final live points, appended in order of increasing logl (dynesty add_final_live)
    order = np.argsort(live_logl)
    niter = len(dead_logl)
    logvol_dead = -np.arange(1, niter + 1) / nlive
    logvol_live = logvol_dead[-1] + np.log1p(-(np.arange(nlive) + 1.0) / (nlive + 1.0))
    samples_x = np.vstack([np.array(dead_x), live[order]])
    logl = np.concatenate([np.array(dead_logl), live_logl[order]])
    logvol = np.concatenate([logvol_dead, logvol_live])
    cum = compute_integrals(logl=logl, logvol=logvol)
    mean = float(cum[-1])
    # information and the sampler's own error estimate sqrt(H / nlive)
    dv = np.diff(logvol, prepend=0)
    logwt = np.logaddexp(np.concatenate([[-1e300], logl])[1:], np.concatenate([[-1e300], logl])[:-1]) + \
        (logvol - dv + np.log1p(-np.exp(dv))) + math.log(0.5)
    w = np.exp(logwt - mean)
    h_info = float(np.sum(w * (logl - mean)))
    logz_err = math.sqrt(max(h_info, 0.0) / nlive)
    success = bool(~np.all(logl == logl[0]))                                  # samplers.py:167

    # GP-uncertainty bounds (samplers.py:172-183): one batched variance call for every sample
    var = np.asarray(gp.predict_var_batched(samples_x), dtype=np.float64)
    std = np.sqrt(var)
    upper = compute_integrals(logl=logl + std, logvol=logvol)
    lower = compute_integrals(logl=logl - std, logvol=logvol)
    var = np.clip(var, 1e-12, 1e12)
    log_var_delta = compute_integrals(logl=2 * logl + np.log(var), logvol=logvol, squared=True)[-1]
    var_logz = math.exp(float(np.clip(log_var_delta - 2 * mean, -100, 100)))
    logz_dict = {"mean": mean, "dlogz_sampler": logz_err, "upper": float(upper[-1]), "lower": float(lower[-1]),
                 "var": var_logz, "std": 2 * math.sqrt(var_logz), "ncall": int(ncall), "niter": int(niter)}
    best_pt = samples_x[int(np.argmax(logl))]
    weights = renormalise_log_weights(logwt)
    if equal_weights:
        samples_x, logl = resample_equal(samples_x, logl, weights, rng=rng)
        weights = np.ones(samples_x.shape[0])
    samples_dict = {"x": samples_x, "weights": weights, "logl": logl, "best": best_pt, "method": "nested"}
    return samples_dict, logz_dict, success
