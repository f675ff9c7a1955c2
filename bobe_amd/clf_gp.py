"""``GPwithClassifier`` — counterpart of ``BOBE/clf_gp.py`` on top of the GPU ``GP``.

All GP arithmetic is inherited (``super()`` calls into libbobe_gp.so, as in the reference, clf_gp.py:173-212);
this class adds what the reference adds: a second, larger data set for a feasibility classifier, the GP
trained only on points within ``gp_threshold`` of the best value (clf_gp.py:86-93, 238-244), and the gating of
the predictions — mean -> ``minus_inf`` and variance -> 1e-12 where the classifier says "infeasible"
(clf_gp.py:173-205).  Only the SVM classifier is provided: scikit-learn's ``SVC`` is TRAINED on the host, as in the
reference (clf.py:36-69); its RBF decision function (clf.py:188-213) and the gate are evaluated ON THE DEVICE
(``bobe_gp_set_gate``: inside ``bobe_gp_predict`` / ``_predict_grad`` / ``_acq_ei`` and the HMC kernels).  This file
holds no classifier arithmetic: it hands the trained parameters to the library and replaces the library's mark for a
gated mean (-inf) by ``minus_inf``.  The Flax MLP / ellipsoid classifiers are optional extras of the reference and are
not built.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .clf import (CLASSIFIER_REGISTRY, _DeviceSVM, gate_eval, get_svm_predict_proba_fn, install_gate,  # noqa: F401
                  train_svm_classifier)
from .gp import GP, safe_noise_floor
from .utils import get_logger, get_numpy_rng

log = get_logger("clf_gp")


class GPwithClassifier(GP):
    def __init__(self, train_x=None, train_y=None, clf_type="svm", clf_settings=None, clf_use_size=10,
                 clf_update_step=1, probability_threshold=0.5, minus_inf=-1e5, clf_threshold=250.0,
                 gp_threshold=500.0, noise=1e-8, kernel="rbf", optimizer="scipy", optimizer_options={},
                 kernel_variance_bounds=[1e-4, 1e8], lengthscale_bounds=[0.01, 5.0], tausq=None,
                 tausq_bounds=[1e-4, 1e4], kernel_variance_prior=None, lengthscale_prior=None, lengthscales=None,
                 kernel_variance=1.0, param_names=None, train_clf_on_init=True, device: int = 0,
                 pivot_floor_ulp: Optional[float] = None):
        """Same keywords as clf_gp.py:15-30 (+ ``device``, ``pivot_floor_ulp``: see ``GP``)."""
        if clf_type.lower() != "svm":
            raise ValueError(f"Unsupported classifier type: {clf_type} (only 'svm' is built)")
        self.train_x_clf = np.array(train_x, dtype=np.float64)
        self.train_y_clf = np.array(train_y, dtype=np.float64).reshape(-1, 1)
        self.clf_use_size, self.clf_update_step = clf_use_size, clf_update_step
        self.clf_type, self.clf_settings = "svm", (clf_settings or {})
        self.clf_params, self.clf_metrics = None, {}
        self.probability_threshold, self.minus_inf = probability_threshold, minus_inf
        self.clf_threshold, self.gp_threshold = clf_threshold, gp_threshold
        mask = self.train_y_clf.flatten() > (self.train_y_clf.max() - self.gp_threshold)       # clf_gp.py:86-89
        super().__init__(train_x=self.train_x_clf[mask], train_y=self.train_y_clf[mask], noise=noise, kernel=kernel,
                         optimizer=optimizer, optimizer_options=optimizer_options,
                         kernel_variance_bounds=kernel_variance_bounds, lengthscale_bounds=lengthscale_bounds,
                         lengthscales=lengthscales, kernel_variance=kernel_variance,
                         lengthscale_prior=lengthscale_prior if lengthscale_prior is not None else "DSLP",
                         kernel_variance_prior=kernel_variance_prior, tausq=tausq, tausq_bounds=tausq_bounds,
                         param_names=param_names, device=device, pivot_floor_ulp=pivot_floor_ulp)
        self._gate_installed = False
        self._clf_predict_func: Optional[Callable] = None
        self.use_clf = self.clf_data_size >= self.clf_use_size
        if self.use_clf and train_clf_on_init:
            self.train_classifier()

    @property
    def clf_data_size(self) -> int:
        return self.train_x_clf.shape[0]

    # ``use_clf`` and the trained parameters decide whether the library gates this GP's predictions: every change of
    # either goes through ``_sync_gate`` (the attribute can be set from outside, as in the reference)
    @property
    def use_clf(self) -> bool:
        return self._use_clf

    @use_clf.setter
    def use_clf(self, value) -> None:
        self._use_clf = bool(value)
        self._sync_gate()

    def _sync_gate(self) -> None:
        """The library gates iff the reference would (clf_gp.py:175-176): ``use_clf`` and a trained classifier."""
        want = bool(getattr(self, "_use_clf", False)) and getattr(self, "clf_params", None) is not None
        if not hasattr(self, "_h") or not self._h.value:
            return                                          # (before GP.__init__ created the handle)
        if want:
            install_gate(self._lib, self._h, self.clf_params, self.probability_threshold, self.minus_inf)
            self._clf_predict_func = self._device_proba
        elif self._gate_installed:
            install_gate(self._lib, self._h, None, self.probability_threshold, self.minus_inf)
        self._gate_installed = want

    def _device_proba(self, x):
        """``svm_predict_proba`` (clf.py:210-213) of the trained classifier, on the device."""
        return gate_eval(self._lib, self._h, x, self.ndim)[1]

    def clf_decision(self, x):
        """``svm_predict`` (clf.py:188-208): the decision values of the trained classifier, on the device."""
        return gate_eval(self._lib, self._h, x, self.ndim)[0]

    def train_classifier(self):
        """clf_gp.py:128-171."""
        if not self.use_clf and self.clf_data_size >= self.clf_use_size:
            self.use_clf = True
        if not self.use_clf:
            return
        labels = np.where(self.train_y_clf.flatten() < self.train_y_clf.max() - self.clf_threshold, 0, 1)
        if np.all(labels == labels[0]):               # one class only: do not use the classifier for the moment
            self.use_clf = False
            return
        self.clf_params, self.clf_metrics, _ = train_svm_classifier(self.train_x_clf, labels, self.clf_settings)
        self._sync_gate()

    def _gated(self) -> bool:
        return self._gate_installed

    # ---- gated predictions (clf_gp.py:173-205): the library marks a gated mean with -inf and returns 1e-12 as its
    # variance; the mark becomes ``minus_inf`` in the units of the method at hand
    def predict_mean_batched(self, x):
        m = super().predict_mean_batched(x)
        return np.where(np.isneginf(m), self.minus_inf, m) if self._gated() else m

    def predict_mean_single(self, x):
        return self.predict_mean_batched(x)[0]

    def predict_var_batched(self, x):
        if not self._gated():
            return super().predict_var_batched(x)
        # (the gate replaces the variance AFTER the y_std^2 scaling, clf_gp.py:186-189: the mean is asked for as well,
        # it carries the gate's mark)
        m, v = self._predict(x, True, True, 0)
        return np.where(np.isneginf(m), safe_noise_floor, self.y_std ** 2 * v)

    def predict_var_single(self, x):
        return self.predict_var_batched(x)[0]

    def predict_batched(self, x):
        m, v = super().predict_batched(x)
        return (np.where(np.isneginf(m), self.minus_inf, m), v) if self._gated() else (m, v)

    def predict_single(self, x):
        m, v = self.predict_batched(x)
        return m[0], v[0:1]

    def predict_grad(self, x, mean_only=False):
        """``GP.predict_grad`` under the gate: a gated point has mean ``minus_inf`` (standardised units, like
        ``predict_single``), variance 1e-12 and zero gradients."""
        m, v, dm, dv = super().predict_grad(x, mean_only=mean_only)
        return (np.where(np.isneginf(m), self.minus_inf, m), v, dm, dv) if self._gated() else (m, v, dm, dv)

    def update(self, new_x, new_y):
        """clf_gp.py:214-246: extend the classifier set, re-derive the GP subset, refactor."""
        new_x = np.atleast_2d(np.asarray(new_x, dtype=np.float64))
        new_y = np.atleast_2d(np.asarray(new_y, dtype=np.float64))
        pts, vals = [], []
        for i in range(new_x.shape[0]):
            if np.any(np.all(np.isclose(self.train_x_clf, new_x[i], atol=1e-6, rtol=1e-4), axis=1)):
                continue
            pts.append(new_x[i])
            vals.append(new_y[i])
        if pts:
            self.train_x_clf = np.concatenate([self.train_x_clf, np.atleast_2d(np.array(pts))], axis=0)
            self.train_y_clf = np.concatenate([self.train_y_clf, np.array(vals).reshape(-1, 1)], axis=0)
            mask = self.train_y_clf.flatten() > (self.train_y_clf.max() - self.gp_threshold)
            self.train_x = self.train_x_clf[mask]
            y = self.train_y_clf[mask].reshape(-1, 1)
            self.y_std = float(np.std(y)) if y.shape[0] > 1 else 1.0
            self.y_mean = float(np.mean(y))
            self.train_y = (y - self.y_mean) / self.y_std
            self._push_data()
            self.recompute_cholesky()

    def kernel(self, x1, x2, lengthscales=None, kernel_variance=None, noise=None, include_noise=True):
        """clf_gp.py:248-252 (argument names of the reference's override)."""
        return super().kernel(x1, x2, lengthscales, kernel_variance, noise, include_noise=include_noise)

    def get_random_point(self, rng=None, nstd=None):
        """clf_gp.py:254-277 (the nstd -> threshold map of utils/core.py is replaced by clf_threshold)."""
        rng = rng if rng is not None else get_numpy_rng()
        if self.use_clf:
            idx = np.where(self.train_y_clf.flatten() > self.train_y_clf.max() - self.clf_threshold)[0]
            return self.train_x_clf[rng.choice(idx, size=1)[0]]
        return super().get_random_point(rng=rng, nstd=nstd)

    @classmethod
    def from_state_dict(cls, state, device: int = 0, _clone_of=None):
        """clf_gp.py:322-386: rebuilt from the CLASSIFIER data set (the GP subset is re-derived by the thresholds),
        hyper-parameters from the state (the GP is factored again on the GPU, as the reference recomputes it),
        classifier parameters and flags restored without retraining."""
        def plain(v):
            return v.item() if isinstance(v, np.ndarray) and v.shape == () else v
        g = cls(train_x=state["train_x_clf"], train_y=state["train_y_clf"], clf_type=plain(state["clf_type"]),
                clf_settings=plain(state["clf_settings"]), clf_use_size=plain(state["clf_use_size"]),
                clf_update_step=plain(state["clf_update_step"]),
                probability_threshold=plain(state["probability_threshold"]), minus_inf=plain(state["minus_inf"]),
                clf_threshold=plain(state["clf_threshold"]), gp_threshold=plain(state["gp_threshold"]),
                noise=plain(state["noise"]), kernel=plain(state["kernel_name"]),
                optimizer=plain(state["optimizer_method"]), optimizer_options=plain(state["optimizer_options"]),
                kernel_variance_bounds=list(np.asarray(state["kernel_variance_bounds"]).tolist()),
                lengthscale_bounds=list(np.asarray(state["lengthscale_bounds"]).tolist()),
                lengthscales=state["lengthscales"], kernel_variance=plain(state["kernel_variance"]),
                kernel_variance_prior=plain(state.get("kernel_variance_prior_spec")),
                lengthscale_prior=plain(state.get("lengthscale_prior_spec")), tausq=plain(state.get("tausq", 1.0)),
                tausq_bounds=list(np.asarray(state.get("tausq_bounds", [1e-4, 1e4])).tolist()),
                train_clf_on_init=False, device=device)
        g.clf_params = plain(state.get("clf_params"))
        g.clf_metrics = plain(state.get("clf_metrics", {})) or {}
        g.use_clf = bool(plain(state["use_clf"]))              # (the setter hands the restored parameters to the library)
        return g

    def state_dict(self, with_factor: bool = True):
        """clf_gp.py:279-320: base GP state + classifier data / configuration / parameters."""
        state = super().state_dict(with_factor=with_factor)
        state.update({"train_x_clf": np.array(self.train_x_clf), "train_y_clf": np.array(self.train_y_clf),
                      "clf_type": self.clf_type, "clf_settings": self.clf_settings, "clf_use_size": self.clf_use_size,
                      "clf_update_step": self.clf_update_step, "probability_threshold": self.probability_threshold,
                      "minus_inf": self.minus_inf, "clf_threshold": self.clf_threshold,
                      "gp_threshold": self.gp_threshold, "use_clf": self.use_clf, "clf_params": self.clf_params,
                      "clf_metrics": self.clf_metrics, "gp_class": "GPwithClassifier"})
        return state
