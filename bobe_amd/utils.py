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
