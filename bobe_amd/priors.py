"""Hyper-parameter priors of the GP (host side, O(d)).

Re-states the NumPyro log-densities the reference uses (BOBE/gp.py:27-78, 309-366) together with
their analytic derivatives, so that ``GP.neg_mll`` can hand SciPy's L-BFGS-B an exact gradient
(the reference gets it from jax.value_and_grad, BOBE/optim.py:306-309).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)
SQRT2 = math.sqrt(2.0)
SQRT3 = math.sqrt(3.0)


class Dist:
    """log_prob(x) and d log_prob / dx, elementwise."""

    def log_prob(self, x):
        raise NotImplementedError

    def dlog_prob(self, x):
        raise NotImplementedError


class Uniform(Dist):
    def __init__(self, low=0.0, high=1.0):
        self.low, self.high = float(low), float(high)

    def log_prob(self, x):  # numpyro: constant, no support masking unless validate_args
        return -math.log(self.high - self.low) * np.ones_like(np.asarray(x, dtype=np.float64))

    def dlog_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))


class Normal(Dist):
    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -0.5 * ((x - self.loc) / self.scale) ** 2 - math.log(self.scale) - 0.5 * LOG_2PI

    def dlog_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -(x - self.loc) / self.scale ** 2


class LogNormal(Dist):
    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        lx = np.log(x)
        return -0.5 * ((lx - self.loc) / self.scale) ** 2 - math.log(self.scale) - 0.5 * LOG_2PI - lx

    def dlog_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -(np.log(x) - self.loc) / (self.scale ** 2 * x) - 1.0 / x


class HalfCauchy(Dist):
    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return math.log(2.0) - math.log(math.pi) - math.log(self.scale) - np.log1p((x / self.scale) ** 2)

    def dlog_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -2.0 * x / (self.scale ** 2 + x * x)


class HalfNormal(Dist):
    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -0.5 * (x / self.scale) ** 2 - math.log(self.scale) - 0.5 * LOG_2PI + math.log(2.0)

    def dlog_prob(self, x):
        return -np.asarray(x, dtype=np.float64) / self.scale ** 2


class Gamma(Dist):
    def __init__(self, concentration=1.0, rate=1.0):
        self.a, self.b = float(concentration), float(rate)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.a * math.log(self.b) + (self.a - 1.0) * np.log(x) - self.b * x - math.lgamma(self.a)

    def dlog_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return (self.a - 1.0) / x - self.b


class Dummy(Dist):
    """BOBE/gp.py:22-25 — fixed kernel variance: log_prob = 0."""

    def log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))

    def dlog_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))


_REGISTRY = {c.__name__: c for c in (Uniform, Normal, LogNormal, HalfCauchy, HalfNormal, Gamma)}


def make_distribution(spec: dict) -> Dist:
    """BOBE/gp.py:27-54 — {'name': ..., **kwargs} -> distribution."""
    cls = _REGISTRY.get(spec["name"])
    if cls is None:
        raise ValueError(f"Distribution {spec['name']} not found (supported: {sorted(_REGISTRY)})")
    return cls(**{k: v for k, v in spec.items() if k != "name"})


def dslp(ndim: int) -> LogNormal:
    """BOBE/gp.py:329-331."""
    return LogNormal(loc=SQRT2 + 0.5 * math.log(ndim), scale=SQRT3)


def saas_logprob_and_grad(lengthscales, kernel_variance, tausq) -> Tuple[float, np.ndarray, float, float]:
    """BOBE/gp.py:56-78 and its derivatives wrt (lengthscales, kernel_variance, tausq)."""
    ls = np.asarray(lengthscales, dtype=np.float64)
    ln, hc01, hc1 = LogNormal(0.0, 1.0), HalfCauchy(0.1), HalfCauchy(1.0)
    u = 1.0 / (tausq * ls ** 2)
    lp = float(ln.log_prob(kernel_variance)) + float(hc01.log_prob(tausq)) + float(np.sum(hc1.log_prob(u)))
    du = hc1.dlog_prob(u)
    g_ls = du * (-2.0 * u / ls)
    g_kvar = float(ln.dlog_prob(kernel_variance))
    g_tau = float(hc01.dlog_prob(tausq)) + float(np.sum(du * (-u / tausq)))
    return lp, g_ls, g_kvar, g_tau
