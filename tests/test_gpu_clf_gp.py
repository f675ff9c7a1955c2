"""GPwithClassifier gating on the GPU GP (reference clf_gp.py:173-205, tests/test_clf_gp.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_data(n=120, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(n, 2))
    y = -2000.0 * np.sum((X - 0.5) ** 2, axis=1, keepdims=True)       # deep well: many points below the thresholds
    return X, y


def test_gating_and_thresholded_training_subset():
    from bobe_amd.clf_gp import GPwithClassifier, get_svm_predict_proba_fn
    from bobe_amd import GP
    X, y = make_data()
    g = GPwithClassifier(X, y, clf_threshold=150.0, gp_threshold=400.0, noise=1e-6, lengthscales=[0.3, 0.3])
    mask = y.ravel() > y.max() - 400.0
    assert g.npoints == int(mask.sum()) < len(y) and g.clf_data_size == len(y)       # clf_gp.py:86-93
    assert g.use_clf and g.clf_metrics["n_support_vectors"] > 0
    q = np.array([[0.5, 0.5], [0.52, 0.47], [0.02, 0.03], [0.97, 0.99]])
    feas = g._clf_predict_func(q) >= 0.5
    assert list(feas) == [True, True, False, False]
    plain = GP(X[mask], y[mask], noise=1e-6, lengthscales=[0.3, 0.3], lengthscale_prior="DSLP")
    m, v = g.predict_mean_batched(q), g.predict_var_batched(q)
    assert np.allclose(m[:2], plain.predict_mean_batched(q[:2]), atol=1e-9) and np.all(m[2:] == g.minus_inf)
    assert np.allclose(v[:2], plain.predict_var_batched(q[:2]), rtol=1e-9) and np.all(v[2:] == 1e-12)
    ms, vs = g.predict_batched(q)
    assert np.all(ms[2:] == g.minus_inf) and np.all(vs[2:] == 1e-12)
    assert g.predict_mean_single(q[2]) == g.minus_inf and g.predict_var_single(q[0]) > 1e-12
    # the SVM decision function restated from the stored parameters (clf.py:188-213) agrees with itself
    f2 = get_svm_predict_proba_fn(g.clf_params)
    assert np.array_equal(f2(X), g._clf_predict_func(X))
    # fantasy variance passes straight through to the GPU GP (clf_gp.py:207-212)
    Z = np.random.default_rng(1).uniform(0.3, 0.7, size=(10, 2))
    assert np.allclose(g.fantasy_var(q[0], Z), plain.fantasy_var(q[0], Z), rtol=1e-9)


def test_update_extends_both_sets_and_state():
    from bobe_amd.clf_gp import GPwithClassifier
    X, y = make_data(60, seed=1)
    g = GPwithClassifier(X, y, clf_threshold=150.0, gp_threshold=400.0, noise=1e-6)
    n_gp, n_clf = g.npoints, g.clf_data_size
    g.update(np.array([[0.5, 0.5], [0.01, 0.01]]), np.array([[0.0], [-960.0]]))
    yall = np.vstack([y, [[0.0], [-960.0]]])
    want = int(np.sum(yall.ravel() > yall.max() - 400.0))                     # subset re-derived from the new max
    assert g.clf_data_size == n_clf + 2 and g.npoints == want                # the far point only feeds the classifier
    g.update(X[:1], y[:1])
    assert g.clf_data_size == n_clf + 2                                      # duplicate rejected
    sd = g.state_dict()
    assert sd["gp_class"] == "GPwithClassifier" and sd["train_x_clf"].shape[0] == n_clf + 2
    pt = g.get_random_point(np.random.default_rng(0))
    assert pt.shape == (2,)
