"""``Likelihood`` - counterpart of the wrapper class in BOBE/likelihood.py:9-124 that ``BOBE`` puts around a plain callable:
a name, the parameter names / LaTeX labels / box bounds, and an evaluation that never raises.  (``CobayaLikelihood``,
likelihood.py:126-, adapts an external sampler framework and is outside the hot path's scope: DESIGN.md 8.)

Same constructor keywords, attributes (``logl, param_list, ndim, param_labels, param_bounds, name, minus_inf,
logprior_vol``) and call convention as the reference class; the code is this build's own.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from .utils.log import get_logger

log = get_logger("likelihood")


def _box(bounds, ndim: int) -> np.ndarray:
    """The (2, ndim) array of lower / upper limits; ``None`` means the unit cube."""
    if bounds is None:
        log.warning("No param_bounds provided. Assuming unit cube [0,1] for all parameters.")
        return np.vstack([np.zeros(ndim), np.ones(ndim)]).astype(int)
    box = np.array(bounds)
    if box.shape != (2, ndim):
        raise ValueError(f"param_bounds must have shape (2, {ndim}), but got {box.shape}.")
    return box


class Likelihood:
    """A log-likelihood together with its parameter space.

    ``loglikelihood(x) -> float`` takes the physical parameter vector.  An evaluation that raises, or returns NaN, an
    infinity or anything below ``minus_inf``, counts as ``minus_inf`` (likelihood.py:61-83): the BO loop never sees an
    exception or a non-finite target."""

    def __init__(self, loglikelihood: Callable, param_list: Optional[List[str]], param_labels: Optional[List[str]] = None,
                 param_bounds: Optional[Union[List, np.ndarray]] = None, name: Optional[str] = None,
                 minus_inf: float = -1e10):
        names: Sequence = param_list
        if any(not isinstance(n, str) for n in names):
            raise ValueError("All elements of param_list must be strings corresponding to parameter names.")
        self.logl = loglikelihood
        self.param_list = param_list
        self.ndim = len(names)
        self.param_labels = ["x_{%d}" % (k + 1) for k in range(self.ndim)] if param_labels is None else param_labels
        self.param_bounds = _box(param_bounds, self.ndim)
        self.name = name if name else "loglikelihood"
        self.minus_inf = minus_inf
        widths = self.param_bounds[1] - self.param_bounds[0]
        self.logprior_vol = np.log(np.prod(widths))                       # (likelihood.py:52: log of the box volume)
        limits = ", ".join("'%s': [%.6g, %.6g]" % (n, lo, hi) for n, lo, hi in zip(names, *self.param_bounds))
        log.info(f"{self.name}: {self.ndim} parameters {{{limits}}}, log prior volume {self.logprior_vol:.4f}")

    def _safe_eval(self, x: np.ndarray) -> float:
        """The value at one parameter vector, ``minus_inf`` for every kind of failure."""
        try:
            value = float(self.logl(x))
        except Exception:                                   # whatever the user's function throws
            log.debug("log-likelihood raised at %s", x, exc_info=True)
            return self.minus_inf
        return value if (math.isfinite(value) and value >= self.minus_inf) else self.minus_inf

    def __call__(self, X: Union[np.ndarray, List[float]]) -> float:
        """One point, given as (ndim,) or (1, ndim) - batches go through the BO driver's loop (likelihood.py:85-124)."""
        point = np.atleast_1d(X)
        if point.ndim > 1:
            if point.shape[0] != 1:
                raise ValueError("__call__ expects a single point. Use pool.run_map_objective for batch evaluations.")
            point = point.reshape(-1)
        if point.shape[0] != self.ndim:
            raise ValueError(f"Input shape {point.shape} does not match ndim {self.ndim}")
        return self._safe_eval(point)
