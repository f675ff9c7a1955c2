"""bobe_amd — MI355X-native GP-surrogate engine behind BOBE's ``GP`` / acquisition surface.

The compute path is libbobe_gp.so (hand-written gfx950 HIP kernels, include/bobe_gp.h); this
package is the thin ctypes layer that presents the reference's Python interface again.
"""
from ._lib import BobeLibraryError, load as load_library  # noqa: F401
from .gp import GP  # noqa: F401
from .acquisition import EI, LogEI, WIPV, WIPStd, get_mc_points, get_mc_samples  # noqa: F401
from .optim import optimize_scipy  # noqa: F401

__all__ = ["GP", "EI", "LogEI", "WIPV", "WIPStd", "get_mc_points", "get_mc_samples", "optimize_scipy",
           "load_library", "BobeLibraryError"]
