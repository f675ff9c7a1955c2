"""Classifier functions - counterpart of the SVM half of BOBE/clf.py (train_svm_classifier clf.py:36-69,
get_svm_predict_proba_fn :71-78, CLASSIFIER_REGISTRY :169-182, svm_predict / svm_predict_proba :188-213).

scikit-learn trains, as in the reference; the decision function is evaluated by the library on the device, by direct
differences in one fixed summation order (``bobe_gp_set_gate`` / ``bobe_gp_gate_eval``, k_gate): a handle that carries
nothing but the gate serves the module-level functions.  The Flax-MLP and ellipsoid classifiers of clf.py:84-166, 221-472
are outside the hot path's scope."""
from __future__ import annotations

from typing import Callable

import numpy as np

from .utils import get_logger

log = get_logger("clf")


def train_svm_classifier(X, Y, settings=None, init_params=None, **kwargs):
    """clf.py:36-69: SVC(kernel='rbf', gamma='scale', C=1e7); returns (params, metrics, predict_proba_fn) - the third
    member evaluates ``svm_predict_proba`` of the fitted parameters on the device.  ``init_params`` / further keywords are
    accepted for the reference's call signature and unused, as there."""
    from sklearn.svm import SVC
    settings = settings or {}
    C = settings.get("C", 1e7)
    clf = SVC(kernel=settings.get("kernel", "rbf"), gamma=settings.get("gamma", "scale"), C=C)
    clf.fit(np.asarray(X), np.asarray(Y))
    params = {"support_vectors": np.array(clf.support_vectors_), "dual_coef": np.array(clf.dual_coef_[0]),
              "intercept": float(clf.intercept_[0]), "gamma_eff": float(clf._gamma)}
    metrics = {"n_support_vectors": len(params["support_vectors"]), "gamma": f"{params['gamma_eff']:.2e}",
               "C": f"{C:.2e}", "intercept": f"{params['intercept']:.2e}"}
    return params, metrics, get_svm_predict_proba_fn(params, device=int(kwargs.get("device", 0)))


def get_svm_predict_proba_fn(params, device: int = 0) -> Callable[[np.ndarray], np.ndarray]:
    """clf.py:71-78: the probability function of stored SVM parameters (``svm_predict_proba``, clf.py:210-213) — evaluated
    by the library (``bobe_gp_gate_eval`` on a data-less handle that carries only the gate)."""
    return _DeviceSVM(params, device).proba


def svm_predict(x, support_vectors, dual_coef, intercept: float, gamma: float):
    """clf.py:188-209: decision(x) = sum_i dual_coef[i] exp(-gamma |support_vectors[i] - x|^2) + intercept, on the device.
    One point (n_features,) gives a scalar, as there; a batch (n, n_features) gives n values."""
    x = np.asarray(x, dtype=np.float64)
    dec = _DeviceSVM({"support_vectors": support_vectors, "dual_coef": dual_coef, "intercept": intercept,
                      "gamma_eff": gamma}).decision(x)
    return float(dec[0]) if x.ndim == 1 else dec


def svm_predict_proba(x, support_vectors, dual_coef, intercept: float, gamma: float):
    """clf.py:211-213: 1.0 where the decision function is >= 0, else 0.0."""
    dec = svm_predict(x, support_vectors, dual_coef, intercept, gamma)
    return np.where(np.asarray(dec) >= 0, 1.0, 0.0) if np.ndim(dec) else (1.0 if dec >= 0 else 0.0)


CLASSIFIER_REGISTRY = {                                   # clf.py:169-182 ('nn' and 'ellipsoid' are not built)
    "svm": {"train_fn": train_svm_classifier, "predict_fn": get_svm_predict_proba_fn},
}


class _DeviceSVM:
    """A library handle holding nothing but a classifier gate: decision values / probabilities of stored parameters."""

    def __init__(self, params, device: int = 0):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self._ndim = int(np.asarray(params["support_vectors"]).shape[1])
        self._h = C.c_void_p(0)
        _lib.check(self._lib.bobe_gp_create(C.byref(self._h), int(device), 0, self._ndim), "bobe_gp_create")
        install_gate(self._lib, self._h, params, 0.5, 0.0)

    def __del__(self):
        try:
            if self._h.value:
                self._lib.bobe_gp_destroy(self._h)
        except Exception:
            pass

    def decision(self, x):
        return gate_eval(self._lib, self._h, x, self._ndim)[0]

    def proba(self, x):
        return gate_eval(self._lib, self._h, x, self._ndim)[1]


def install_gate(lib, handle, params, probability_threshold: float, minus_inf: float) -> None:
    """Hand the trained SVM to the library (``bobe_gp_set_gate``); ``params=None`` clears the gate."""
    from . import _lib
    if params is None:
        _lib.check(lib.bobe_gp_set_gate(handle, None, 0, None, 0.0, 0.0, float(probability_threshold), float(minus_inf)),
                   "bobe_gp_set_gate")
        return
    sv = _lib.as_f64(np.atleast_2d(np.asarray(params["support_vectors"])))
    dual = _lib.as_f64(np.asarray(params["dual_coef"])).reshape(-1)
    _lib.check(lib.bobe_gp_set_gate(handle, _lib.ptr(sv), sv.shape[0], _lib.ptr(dual), float(params["intercept"]),
                                    float(params["gamma_eff"]), float(probability_threshold), float(minus_inf)),
               "bobe_gp_set_gate")


def gate_eval(lib, handle, x, ndim: int):
    """(decision, feasible) of the points ``x`` from the gate held by ``handle`` (``bobe_gp_gate_eval``)."""
    from . import _lib
    x = _lib.as_f64(np.atleast_2d(np.asarray(x, dtype=np.float64)).reshape(-1, ndim))
    dec, ok = np.empty(x.shape[0]), np.empty(x.shape[0])
    _lib.check(lib.bobe_gp_gate_eval(handle, _lib.ptr(x), x.shape[0], _lib.ptr(dec), _lib.ptr(ok)), "bobe_gp_gate_eval")
    return dec, ok
