"""Multi-restart L-BFGS-B driver — counterpart of BOBE/optim.py:249-359 (``optimize_scipy``).

The reference wraps ``jax.jit(jax.value_and_grad(fun))`` (optim.py:306-309); here the caller hands in
a ``value_and_grad(x) -> (f, g)`` callable whose heavy part runs on the GPU through the C ABI.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
from scipy.optimize import minimize

from .utils import get_logger

log = get_logger("optim")


def _setup_bounds(bounds, num_params):
    """BOBE/optim.py:42-68."""
    if bounds is None:
        return None
    bounds = np.array(bounds, dtype=np.float64)
    if bounds.shape == (2,):
        bounds = np.tile(bounds.reshape(1, 2), (num_params, 1)).T
    elif bounds.shape != (2, num_params):
        raise ValueError(f"Bounds shape {bounds.shape} incompatible with {num_params} parameters")
    return bounds


def optimize_scipy(value_and_grad: Callable, num_params: int = 1, bounds=None, x0=None,
                   optimizer_options: Optional[dict] = None, maxiter: int = 200, n_restarts: int = 4,
                   verbose: bool = False) -> Tuple[np.ndarray, float]:
    """Same restart / screening / acceptance logic as BOBE/optim.py:292-359.

    ``value_and_grad`` may return ``g=None``; SciPy then falls back to finite differences.
    """
    options = dict(optimizer_options if optimizer_options is not None else
                   {"method": "L-BFGS-B", "ftol": 1e-6, "gtol": 1e-6})      # optim.py:256
    options.update({"maxiter": maxiter})                                   # optim.py:292
    method = options.pop("method", "L-BFGS-B")                             # optim.py:294
    bounds_arr = _setup_bounds(bounds, num_params)
    scipy_bounds = None if bounds_arr is None else [
        (float(bounds_arr[0, i]), float(bounds_arr[1, i])) for i in range(num_params)]
    if x0 is None:
        raise ValueError("x0 must be provided (shape: (n_restarts, num_params) or (num_params,))")
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    if x0.shape[0] < n_restarts:
        raise ValueError(f"x0 provided with {x0.shape[0]} restarts but n_restarts={n_restarts}")
    x0 = x0[:n_restarts]

    probe = value_and_grad(x0[0])
    has_grad = probe[1] is not None
    fun = value_and_grad if has_grad else (lambda x: value_and_grad(x)[0])

    best_f, best_x = np.inf, None
    for i, x_init in enumerate(x0):                                        # optim.py:325-333
        try:
            val = probe[0] if i == 0 else value_and_grad(x_init)[0]
            if np.isfinite(val) and val < best_f:
                best_f, best_x = float(val), np.array(x_init)
        except Exception as e:  # pragma: no cover
            log.warning(f"  Initial point {i + 1}/{n_restarts}: failed with {e}")
    for i, x_init in enumerate(x0):                                        # optim.py:335-354
        try:
            res = minimize(fun, x_init, method=method, jac=has_grad, bounds=scipy_bounds, options=options)
        except Exception as e:
            if verbose:
                log.warning(f"  Restart {i + 1}/{n_restarts}: failed with {e}")
            continue
        ok = res.success or "ITERATIONS REACHED LIMIT" in str(res.message).upper()   # optim.py:340
        if ok and np.isfinite(res.fun) and res.fun < best_f:
            best_f, best_x = float(res.fun), np.array(res.x)
    if best_x is None:
        best_x = np.array(x0[0])
    return np.asarray(best_x), float(best_f)
