"""ctypes binding of oracle/libbobe_oracle_c.so (the plain-C second restatement; TEST INFRASTRUCTURE).

Built by ``make -C oracle`` (``__graft_entry__.build()`` does it).  Only tests may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BOBE_ORACLE_C_LIB selects another build of the same source (the sanitizer build of `make -C oracle asan`)
_PATH = os.environ.get("BOBE_ORACLE_C_LIB") or os.path.join(_HERE, "libbobe_oracle_c.so")
_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HERE], check=True)
        _lib = C.CDLL(_PATH)
        _lib.oc_mll.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def kernel(kern, A, B, ls, kvar, noise, include_noise):
    A, B, ls = _f(np.atleast_2d(A)), _f(np.atleast_2d(B)), _f(ls)
    K = np.empty((A.shape[0], B.shape[0]))
    load().oc_kernel(C.c_int(kern), A.ctypes.data_as(C.c_void_p), C.c_int(A.shape[0]), B.ctypes.data_as(C.c_void_p),
                     C.c_int(B.shape[0]), C.c_int(A.shape[1]), ls.ctypes.data_as(C.c_void_p), C.c_double(kvar),
                     C.c_double(noise), C.c_int(int(include_noise)), K.ctypes.data_as(C.c_void_p))
    return K


def mll(kern, X, y, ls, kvar, noise, want_grad=True):
    """-> (info, mll, grad | None, L, alpha) for standardised targets y."""
    X, y, ls = _f(X), _f(y).reshape(-1), _f(ls)
    n, d = X.shape
    out = C.c_double()
    grad = np.empty(d + 1) if want_grad else None
    L, alpha = np.empty((n, n)), np.empty(n)
    info = load().oc_mll(C.c_int(kern), X.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int(n),
                         C.c_int(d), ls.ctypes.data_as(C.c_void_p), C.c_double(kvar), C.c_double(noise),
                         C.byref(out), grad.ctypes.data_as(C.c_void_p) if want_grad else None,
                         L.ctypes.data_as(C.c_void_p), alpha.ctypes.data_as(C.c_void_p))
    return info, out.value, grad, L, alpha


def predict(kern, X, L, alpha, ls, kvar, noise, Xq):
    X, L, alpha, ls, Xq = _f(X), _f(L), _f(alpha), _f(ls), _f(np.atleast_2d(Xq))
    mean, var = np.empty(Xq.shape[0]), np.empty(Xq.shape[0])
    load().oc_predict(C.c_int(kern), X.ctypes.data_as(C.c_void_p), C.c_int(X.shape[0]), C.c_int(X.shape[1]),
                      L.ctypes.data_as(C.c_void_p), alpha.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p),
                      C.c_double(kvar), C.c_double(noise), Xq.ctypes.data_as(C.c_void_p), C.c_int(Xq.shape[0]),
                      mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p))
    return mean, var


def fantasy_var(kern, X, L, ls, kvar, noise, xnew, Z, y_std):
    X, L, ls, xnew, Z = _f(X), _f(L), _f(ls), _f(xnew).reshape(-1), _f(np.atleast_2d(Z))
    out = np.empty(Z.shape[0])
    load().oc_fantasy_var(C.c_int(kern), X.ctypes.data_as(C.c_void_p), C.c_int(X.shape[0]), C.c_int(X.shape[1]),
                          L.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p), C.c_double(kvar),
                          C.c_double(noise), xnew.ctypes.data_as(C.c_void_p), Z.ctypes.data_as(C.c_void_p),
                          C.c_int(Z.shape[0]), C.c_double(y_std), out.ctypes.data_as(C.c_void_p))
    return out


# ---- extended-precision truth (oracle/bobe_oracle_xp.c): long double ("xp") or __float128 ("xq") --------------------
_xlibs = {}


def load_xp(kind: str = "xp"):
    if kind not in _xlibs:
        # BOBE_ORACLE_XP_LIB selects another build of the long-double library (the sanitizer build of `make -C oracle asan`)
        path = (kind == "xp" and os.environ.get("BOBE_ORACLE_XP_LIB")) or os.path.join(_HERE, f"libbobe_oracle_{kind}.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, os.path.basename(path)], check=True)
        lib = C.CDLL(path)
        lib.xp_gp_truth.restype = C.c_int
        lib.xp_digits.restype = C.c_int
        _xlibs[kind] = lib
    return _xlibs[kind]


def gp_truth(kern, X, y, ls, kvar, noise, Xq=None, Z=None, want_grad=True, kind: str = "xp"):
    """The hot path's quantities in extended precision from fp64 inputs (y: standardised targets).  Returns a dict:
    info (0, or 1 + the column of the first non-positive pivot), mll, grad, mean, var, var_z, fantasy (c x m), min_pivot,
    digits - variances in standardised units, noise included, no floors."""
    lib = load_xp(kind)
    X, y, ls = _f(X), _f(y).reshape(-1), _f(ls)
    n, d = X.shape
    Xq = _f(np.atleast_2d(Xq)) if Xq is not None else np.empty((0, d))
    Z = _f(np.atleast_2d(Z)) if Z is not None else np.empty((0, d))
    c, m = Xq.shape[0], Z.shape[0]
    mll, mp = C.c_double(), C.c_double()
    grad = np.empty(d + 1) if want_grad else None
    mean, var, var_z, fant, cross = np.empty(c), np.empty(c), np.empty(m), np.empty((c, m)), np.empty((c, m))

    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None
    info = lib.xp_gp_truth(C.c_int(kern), p(X), p(y), C.c_int(n), C.c_int(d), p(ls), C.c_double(kvar), C.c_double(noise),
                           p(Xq), C.c_int(c), p(Z), C.c_int(m), C.byref(mll), p(grad), p(mean), p(var), p(var_z),
                           p(fant) if c and m else None, C.byref(mp), p(cross) if c and m else None)
    if info != 0:
        return {"info": int(info), "digits": int(lib.xp_digits())}
    return {"info": 0, "mll": mll.value, "grad": grad, "mean": mean, "var": var, "var_z": var_z, "fantasy": fant, "cross": cross,
            "min_pivot": mp.value, "digits": int(lib.xp_digits())}
