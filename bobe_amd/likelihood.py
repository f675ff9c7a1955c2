"""``Likelihood`` - counterpart of BOBE/likelihood.py:9-124: the wrapper ``BOBE`` puts around a plain callable (name,
parameter names / labels / bounds, safe evaluation).  ``CobayaLikelihood`` (likelihood.py:126-) is an adaptor to an
external sampler framework and outside the hot path's scope."""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import numpy as np

from .utils.log import get_logger

log = get_logger("likelihood")


class Likelihood:
    """A log-likelihood with its parameter space.  ``loglikelihood(x) -> float`` on physical parameters; ``param_bounds`` has
    shape (2, ndim) (default: the unit cube); a failed, NaN, infinite or below-``minus_inf`` evaluation returns
    ``minus_inf`` (likelihood.py:61-83)."""

    def __init__(self, loglikelihood: Callable, param_list: Optional[List[str]], param_labels: Optional[List[str]] = None,
                 param_bounds: Optional[Union[List, np.ndarray]] = None, name: Optional[str] = None,
                 minus_inf: float = -1e10):
        self.logl = loglikelihood
        if not all(isinstance(p, str) for p in param_list):
            raise ValueError("All elements of param_list must be strings corresponding to parameter names.")
        self.param_list = param_list
        self.ndim = len(self.param_list)
        self.param_labels = param_labels if param_labels is not None else [f"x_{{{i + 1}}}" for i in range(self.ndim)]
        if param_bounds is None:
            self.param_bounds = np.array(self.ndim * [[0, 1]]).T
            log.warning("No param_bounds provided. Assuming unit cube [0,1] for all parameters.")
        else:
            param_bounds = np.array(param_bounds)
            if param_bounds.shape != (2, self.ndim):
                raise ValueError(f"param_bounds must have shape (2, {self.ndim}), but got {param_bounds.shape}.")
            self.param_bounds = param_bounds
        self.name = name or "loglikelihood"
        self.minus_inf = minus_inf
        self.logprior_vol = np.log(np.prod(self.param_bounds[1] - self.param_bounds[0]))
        log.info(f"Initialized {self.name} with {self.ndim} params: {self.param_list}; log prior volume = "
                 f"{self.logprior_vol:.4f}")

    def _safe_eval(self, x: np.ndarray) -> float:
        try:
            val = float(self.logl(x))
        except Exception:
            log.debug(f"Log-likelihood evaluation failed at point {x}", exc_info=True)
            return self.minus_inf
        if np.isnan(val) or np.isinf(val) or val < self.minus_inf:
            return self.minus_inf
        return val

    def __call__(self, X: Union[np.ndarray, List[float]]) -> float:
        """One point, shape (ndim,) or (1, ndim) (likelihood.py:85-124)."""
        X = np.atleast_1d(X)
        if X.ndim > 1:
            if X.shape[0] != 1:
                raise ValueError("__call__ expects a single point.")
            X = X.flatten()
        if X.shape[0] != self.ndim:
            raise ValueError(f"Input shape {X.shape} does not match ndim {self.ndim}")
        return self._safe_eval(X)
