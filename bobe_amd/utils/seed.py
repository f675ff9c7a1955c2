"""Global seeding - counterpart of BOBE/utils/seed.py (the NumPy half; there are no JAX keys in this build: the device
samplers take integer seeds drawn from this generator)."""
from __future__ import annotations

import os
import random
from typing import Optional

import numpy as np

_global_seed: Optional[int] = None
_rng: Optional[np.random.Generator] = None


def set_global_seed(seed: Optional[int] = None) -> int:
    """BOBE/utils/seed.py:26-50: seed Python's and NumPy's global generators; returns the seed used."""
    global _global_seed, _rng
    if seed is None:
        seed = random.randint(0, 2 ** 31 - 1)
    elif not isinstance(seed, (int, np.integer)) or seed < 0:
        raise ValueError("Seed must be a non-negative integer or None")
    _global_seed = int(seed)
    random.seed(_global_seed)
    _rng = np.random.default_rng(_global_seed)
    os.environ["PYTHONHASHSEED"] = str(_global_seed)
    return _global_seed


def get_global_seed() -> int:
    """BOBE/utils/seed.py:53-62."""
    if _global_seed is None:
        set_global_seed()
    return _global_seed


def get_numpy_rng() -> np.random.Generator:
    """BOBE/utils/seed.py:103-112: the global generator (initialised with a random seed on first use)."""
    if _rng is None:
        set_global_seed()
    return _rng


def ensure_reproducibility(seed: Optional[int] = None) -> int:
    """BOBE/utils/seed.py:114-: set the seeds (fp64 is the only arithmetic here, there is no x64 switch to flip)."""
    return set_global_seed(seed)
