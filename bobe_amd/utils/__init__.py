"""Utilities package - counterpart of BOBE/utils (``core``, ``log``, ``seed``; the results manager and the plotting
helpers of the reference's package are outside the hot path's scope)."""
from .core import (get_threshold_for_nsigma, is_cluster_environment, kl_divergence_gaussian, kl_divergence_samples,  # noqa: F401
                   renormalise_log_weights, resample_equal, scale_from_unit, scale_to_unit, suppress_stdout_stderr)
from .log import get_logger, setup_logging, update_verbosity  # noqa: F401
from .seed import ensure_reproducibility, get_global_seed, get_numpy_rng, set_global_seed  # noqa: F401

__all__ = ["suppress_stdout_stderr", "scale_to_unit", "scale_from_unit", "renormalise_log_weights", "resample_equal",
           "is_cluster_environment", "get_logger", "setup_logging", "get_numpy_rng", "set_global_seed",
           "kl_divergence_gaussian", "get_threshold_for_nsigma", "update_verbosity", "get_global_seed",
           "ensure_reproducibility"]
