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


def get_threshold_for_nsigma(nsigma: float, d: int) -> float:
    """Log-probability drop from the peak of a d-dimensional Gaussian to its n-sigma contour (utils/core.py:150-167)."""
    from scipy.special import erfc
    from scipy.stats import chi2
    nstd = np.sqrt(chi2.isf(erfc(nsigma / np.sqrt(2.0)), d))
    return float(0.5 * nstd ** 2)
