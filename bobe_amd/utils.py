"""Small host helpers (logging, global RNG, unit-cube scaling) — counterparts of
BOBE/utils/log.py, BOBE/utils/seed.py and BOBE/utils/core.py:181-193."""
from __future__ import annotations

import logging

import numpy as np

_rng = np.random.default_rng()


def get_logger(name: str) -> logging.Logger:
    return logging.getLogger(f"bobe_amd.{name}")


def set_global_seed(seed: int) -> None:
    global _rng
    _rng = np.random.default_rng(seed)


def get_numpy_rng() -> np.random.Generator:
    return _rng


def scale_to_unit(x, param_bounds):
    """BOBE/utils/core.py:181-186."""
    return (x - param_bounds[0]) / (param_bounds[1] - param_bounds[0])


def scale_from_unit(x, param_bounds):
    """BOBE/utils/core.py:188-193."""
    return x * (param_bounds[1] - param_bounds[0]) + param_bounds[0]


def renormalise_log_weights(logw):
    """BOBE/utils/core.py counterpart (the reference's examples import it from there): exp(logw - logsumexp(logw))."""
    logw = np.asarray(logw, dtype=np.float64)
    w = np.exp(logw - np.max(logw))
    return w / np.sum(w)


def get_threshold_for_nsigma(nsigma: float, d: int) -> float:
    """Log-probability drop from the peak of a d-dimensional Gaussian to its n-sigma contour (utils/core.py:150-167)."""
    from scipy.special import erfc
    from scipy.stats import chi2
    nstd = np.sqrt(chi2.isf(erfc(nsigma / np.sqrt(2.0)), d))
    return float(0.5 * nstd ** 2)


def _kl_gaussian_single(mu1, cov1, mu2, cov2) -> float:
    """KL(N(mu1, cov1) || N(mu2, cov2)) = 0.5 [tr(S2^-1 S1) + (mu2-mu1)^T S2^-1 (mu2-mu1) - d + ln det S2 - ln det S1]."""
    mu1, mu2 = np.atleast_1d(mu1).astype(float), np.atleast_1d(mu2).astype(float)
    cov1, cov2 = np.atleast_2d(cov1).astype(float), np.atleast_2d(cov2).astype(float)
    d = mu1.shape[0]
    sol = np.linalg.solve(cov2, np.column_stack([cov1, mu2 - mu1]))
    _, ld1 = np.linalg.slogdet(cov1)
    _, ld2 = np.linalg.slogdet(cov2)
    return float(0.5 * (np.trace(sol[:, :d]) + (mu2 - mu1) @ sol[:, d] - d + ld2 - ld1))


def kl_divergence_gaussian(mu1, Cov1, mu2, Cov2) -> dict:
    """Forward, reverse and symmetric KL divergence between two multivariate normals (utils/core.py:132-145: the
    bookkeeping ``check_convergence_logz`` keeps beside the logZ test, bo.py:896-911)."""
    fwd = _kl_gaussian_single(mu1, Cov1, mu2, Cov2)
    rev = _kl_gaussian_single(mu2, Cov2, mu1, Cov1)
    return {"forward": fwd, "reverse": rev, "symmetric": 0.5 * (fwd + rev)}
