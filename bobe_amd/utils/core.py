"""Small host helpers - counterpart of BOBE/utils/core.py (unit-cube scaling, weight handling, Gaussian KL divergence, the
classifier threshold), same names and arguments."""
from __future__ import annotations

import contextlib
import os

import numpy as np

from .seed import get_numpy_rng


def is_cluster_environment() -> bool:
    """BOBE/utils/core.py:15-47: a batch scheduler's job id in the environment, or no terminal on stdout."""
    if any(v in os.environ for v in ("SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID", "SGE_TASK_ID")):
        return True
    try:
        return not os.isatty(1)
    except OSError:
        return True


@contextlib.contextmanager
def suppress_stdout_stderr():
    """BOBE/utils/core.py:197-: silence both streams inside the block."""
    with open(os.devnull, "w") as null, contextlib.redirect_stdout(null), contextlib.redirect_stderr(null):
        yield


def resample_equal(samples, aux, weights=None, logwts=None, *, rng=None):
    """BOBE/utils/core.py:54-77 (taken from jaxns there): systematic resampling of weighted samples to equal weights,
    then a random permutation; ``aux`` rides along.  ``rng`` (keyword-only, this build's): the generator to draw from,
    default the global one as in the reference."""
    rstate = rng if rng is not None else get_numpy_rng()
    wts = renormalise_log_weights(logwts) if logwts is not None else np.asarray(weights, dtype=np.float64)
    cum = np.cumsum(wts / wts.sum())
    cum /= cum[-1]
    n = len(wts)
    positions = (rstate.random() + np.arange(n)) / n
    idx = np.minimum(np.searchsorted(cum, positions, side="right"), n - 1)     # first j with positions[i] < cum[j]
    perm = rstate.permutation(n)
    return np.asarray(samples)[idx][perm], np.asarray(aux)[idx][perm]


def scale_to_unit(x, param_bounds):
    """BOBE/utils/core.py:181-186."""
    return (x - param_bounds[0]) / (param_bounds[1] - param_bounds[0])


def scale_from_unit(x, param_bounds):
    """BOBE/utils/core.py:188-193."""
    return x * (param_bounds[1] - param_bounds[0]) + param_bounds[0]


def renormalise_log_weights(log_weights):
    """BOBE/utils/core.py:49-52 (the reference's examples import it from there): exp(log_weights - logsumexp(log_weights))."""
    logw = np.asarray(log_weights, dtype=np.float64)
    w = np.exp(logw - np.max(logw))
    return w / np.sum(w)


def get_threshold_for_nsigma(nsigma: float, d: int) -> float:
    """Log-probability drop from the peak of a d-dimensional Gaussian to its n-sigma contour (utils/core.py:150-167)."""
    from scipy.special import erfc
    from scipy.stats import chi2
    nstd = np.sqrt(chi2.isf(erfc(nsigma / np.sqrt(2.0)), d))
    return float(0.5 * nstd ** 2)


def kl_divergence_samples(prev_loglike, curr_loglike) -> dict:
    """BOBE/utils/core.py:82-105: forward / reverse / symmetric KL divergence between the normalised likelihood weights of
    two successive iterations' samples."""
    from scipy import stats
    prev = np.asarray(prev_loglike, dtype=np.float64)
    curr = np.asarray(curr_loglike, dtype=np.float64)
    p_prev, p_curr = np.exp(prev - np.max(prev)), np.exp(curr - np.max(curr))
    p_prev, p_curr = p_prev / np.sum(p_prev), p_curr / np.sum(p_curr)
    fwd, rev = stats.entropy(p_prev, p_curr), stats.entropy(p_curr, p_prev)
    return {"forward": fwd, "reverse": rev, "symmetric": 0.5 * (fwd + rev)}


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
