"""``import BOBE`` resolving to ``bobe_amd`` - the import-path alias for scripts written against the reference.

Put this directory's parent (``bobe_amd/compat``) on ``sys.path`` / ``PYTHONPATH`` and an unmodified script's
``from BOBE import BOBE``, ``from BOBE.gp import GP``, ``from BOBE.utils.core import scale_from_unit`` ... bind to the
MI355X-backed counterparts.  Only the modules on the hot path and its callers exist
(DESIGN.md 8): ``BOBE.utils.results`` / ``BOBE.utils.plot`` and the Cobaya adaptor do not, and say so on import.
"""
import importlib
import sys

import bobe_amd as _pkg
from bobe_amd import *  # noqa: F401,F403
from bobe_amd import __all__ as _all

__all__ = list(_all)
__version__ = getattr(_pkg, "__version__", "0.2.0")

for _name in ("gp", "bo", "acquisition", "clf", "clf_gp", "samplers", "optim", "likelihood", "utils", "utils.core",
              "utils.log", "utils.seed"):
    _mod = importlib.import_module("bobe_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        globals()[_name] = _mod


def __getattr__(name):
    if name == "GPwithClassifier":
        return _pkg.GPwithClassifier
    if name in ("BOBEResults", "BOBESummaryPlotter", "CobayaLikelihood"):
        raise AttributeError(f"BOBE.{name} belongs to the parts of the reference that bobe_amd does not build (results manager, "
                             "plotting, Cobaya adaptor); see DESIGN.md section 8")
    raise AttributeError(name)
