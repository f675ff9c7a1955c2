"""bobe_amd — MI355X-native GP-surrogate engine behind BOBE's ``GP`` / acquisition surface.

The compute path is libbobe_gp.so (hand-written gfx950 HIP kernels, include/bobe_gp.h); this
package is the thin ctypes layer that presents the reference's Python interface again.
"""
from ._lib import BobeLibraryError, load as load_library  # noqa: F401
from .gp import GP  # noqa: F401
from .acquisition import EI, LogEI, WIPV, WIPStd, get_mc_points, get_mc_samples  # noqa: F401
from .optim import optimize_scipy  # noqa: F401
from .bo import BOBE, gp_fit  # noqa: F401
from .samplers import compute_integrals, nested_sampling  # noqa: F401


def __getattr__(name):            # scikit-learn is only needed for the classifier GP
    if name == "GPwithClassifier":
        from .clf_gp import GPwithClassifier
        return GPwithClassifier
    raise AttributeError(name)


__all__ = ["GP", "GPwithClassifier", "EI", "LogEI", "WIPV", "WIPStd", "get_mc_points", "get_mc_samples",
           "optimize_scipy", "BOBE", "gp_fit", "nested_sampling", "compute_integrals", "load_library",
           "BobeLibraryError"]
