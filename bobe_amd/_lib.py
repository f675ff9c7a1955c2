"""ctypes binding of libbobe_gp.so (include/bobe_gp.h).

There is no CPU fallback: if the shared library is missing or no HIP device is visible,
the first use raises ``BobeLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbobe_gp.so")

BOBE_OK = 0
BOBE_NOT_PD = 1
MAX_MLL_SLOTS = 8
PROF = {"potf2": 1, "trsm": 2, "syrk": 3, "trtri": 4, "lauum": 5, "trimul": 6, "cross": 7, "kxx": 8, "crossvv": 9, "kxc": 10}

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class BobeLibraryError(RuntimeError):
    pass


# (name, restype, argtypes) — must list every symbol include/bobe_gp.h declares
SIGNATURES = [
    ("bobe_version", C.c_char_p, []),
    ("bobe_last_error", C.c_char_p, []),
    ("bobe_device_count", C.c_int, []),
    ("bobe_gp_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    ("bobe_gp_destroy", None, [C.c_void_p]),
    ("bobe_gp_get_stream", C.c_void_p, [C.c_void_p]),
    ("bobe_gp_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bobe_gp_sync", C.c_int, [C.c_void_p]),
    ("bobe_gp_set_data", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    ("bobe_gp_set_hyper", C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
    ("bobe_gp_factor", C.c_int, [C.c_void_p]),
    ("bobe_gp_mll", C.c_int, [C.c_void_p, C.c_void_p, C.c_double, c_double_p, C.c_void_p]),
    ("bobe_gp_mll_batch", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_mll_submit", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int]),
    ("bobe_gp_mll_wait", C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_void_p]),
    ("bobe_gp_predict", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]),
    ("bobe_gp_wip_sweep", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    c_int64_p, c_double_p, c_int64_p, c_double_p]),
    ("bobe_gp_fantasy_var", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
    ("bobe_gp_wip_grad", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_acq_ei", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    ("bobe_gp_predict_grad", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    ("bobe_gp_hmc_leapfrog", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                       C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_hmc_run", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int,
                                  C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    ("bobe_gp_rwalk", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_uint64,
                                C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_set_gate", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                   C.c_double]),
    ("bobe_gp_gate_eval", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("bobe_gp_kernel", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                 C.c_double, C.c_int, C.c_void_p]),
    ("bobe_gp_dist_sq", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    ("bobe_gp_mll_from_k", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, c_double_p]),
    ("bobe_gp_chol_row_update", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                                          c_double_p]),
    ("bobe_gp_set_pivot_floor_ulp", C.c_int, [C.c_void_p, C.c_double]),
    ("bobe_gp_get_pivot_floor_ulp", C.c_double, [C.c_void_p]),
    ("bobe_gp_set_refine_kappa", C.c_int, [C.c_void_p, C.c_double]),
    ("bobe_gp_set_solve_block", C.c_int, [C.c_void_p, C.c_int]),
    ("bobe_gp_get_solve_block", C.c_int, [C.c_void_p]),
    ("bobe_debug_solve_opts", C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    ("bobe_gp_get_refine", C.c_int, [C.c_void_p, c_double_p, C.POINTER(C.c_int)]),
    ("bobe_gp_get_chol", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_set_chol", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bobe_gp_clone_state", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bobe_gp_append", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("bobe_mgpu_unique_id", C.c_int, [C.c_char_p]),
    ("bobe_mgpu_init", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int]),
    ("bobe_mgpu_world", C.c_int, []),
    ("bobe_mgpu_rank", C.c_int, []),
    ("bobe_mgpu_finalize", None, []),
    ("bobe_mgpu_wip_sweep", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      c_int64_p, c_double_p, c_int64_p, c_double_p]),
    ("bobe_mgpu_best_fit", C.c_int, [C.c_double, C.c_void_p, C.c_int, c_double_p, C.c_void_p]),
    ("bobe_gp_npoints", C.c_int64, [C.c_void_p]),
    ("bobe_debug_gemm", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    ("bobe_debug_kinv", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bobe_debug_linv", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bobe_debug_time_potrf", C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    ("bobe_debug_time_potrf_batch", C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    ("bobe_debug_time_potrf_lockstep", C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    ("bobe_gp_set_chunk", C.c_int, [C.c_void_p, C.c_int64]),
    ("bobe_gp_profile_select", C.c_int, [C.c_void_p, C.c_int]),
    ("bobe_gp_profile_read", C.c_int, [C.c_void_p, c_double_p, c_int64_p]),
    ("bobe_debug_mfma_peak", C.c_int, [C.c_int, C.c_int, c_double_p]),
    ("bobe_debug_wave_sums", C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
]

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libbobe_gp.so and declare every entry point (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BobeLibraryError(
            f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bobe_amd/csrc`. bobe_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().bobe_last_error().decode("utf-8", "replace")


def check(status: int, what: str) -> int:
    """Raise on usage / HIP errors (< 0); numerical conditions (> 0) are returned to the caller."""
    if status < 0:
        raise BobeLibraryError(f"{what} failed ({status}): {last_error()}")
    return status


def ptr(a) -> C.c_void_p:
    """Raw pointer of a C-contiguous float64/int64 NumPy array or a torch tensor (host or device)."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):  # torch tensor
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"unsupported buffer type {type(a)}")


def as_f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        out = out.reshape(shape)
    return out
