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
from .likelihood import Likelihood  # noqa: F401
from .utils import get_logger, scale_from_unit, scale_to_unit, setup_logging  # noqa: F401


def __getattr__(name):            # scikit-learn is only needed for the classifier GP
    if name == "GPwithClassifier":
        from .clf_gp import GPwithClassifier
        return GPwithClassifier
    raise AttributeError(name)


# (the names of the reference's BOBE/__init__.py:70-91 that lie on the path - its results manager and plotter do not -
# then this package's own)
__all__ = ["BOBE", "GP", "GPwithClassifier", "Likelihood", "EI", "LogEI", "WIPV", "WIPStd", "get_logger", "setup_logging",
           "scale_to_unit", "scale_from_unit", "get_mc_points", "get_mc_samples",
           "optimize_scipy", "gp_fit", "nested_sampling", "compute_integrals", "load_library",
           "BobeLibraryError"]
