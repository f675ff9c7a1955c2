"""The reference's own test expectations for the hot path, restated against the MI355X engine.

One test here per test of the reference's ``tests/test_gp.py``, ``tests/test_acquisition.py`` and the SVM tests of
``tests/test_clf_gp.py`` (plus the two GP-related ones of ``tests/test_mpi.py``): same synthetic data recipe, same calls on the same method surface, same
assertions — cited as ``file:line`` of the reference test.  Nothing numeric is pinned by those tests,
they pin the surface and the invariants a drop-in must keep; the numeric parity lives in test_gpu_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _GP(*a, **k):
    from bobe_amd import GP
    return GP(*a, **k)


def quadratic_data(n_samples=50, d=2, seed=42, centre=0.5):
    """tests/test_gp.py:21-27 (centre 0.5) and tests/test_acquisition.py:21-26 (centre 0.7)."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(n_samples, d))
    y = -np.sum((X - centre) ** 2, axis=1).reshape(-1, 1)
    return X, y


def acquisition_gp(n_samples=30, d=2, seed=42):
    """tests/test_acquisition.py:21-37."""
    X, y = quadratic_data(n_samples, d, seed, centre=0.7)
    return _GP(train_x=X, train_y=y, noise=1e-6, kernel="rbf", lengthscales=np.array([0.3] * d), kernel_variance=1.0)


# ---------------------------------------------------------------------------------------- tests/test_gp.py
def test_gp_initialization():
    """tests/test_gp.py:30-55."""
    X, y = quadratic_data(20, 3)
    gp = _GP(train_x=X, train_y=y, noise=1e-6, kernel="rbf", lengthscale_bounds=[0.01, 10],
             kernel_variance_bounds=[1e-4, 1e4])
    assert gp.ndim == 3
    assert gp.train_x.shape[0] == 20 and gp.npoints == 20
    assert gp.kernel_name == "rbf"
    assert np.isfinite(gp.y_mean) and gp.y_std > 0


def test_gp_fitting():
    """tests/test_gp.py:58-89: Matérn kernel, DSLP prior, fit from the current hyper-parameters."""
    X, y = quadratic_data(30, 2)
    gp = _GP(train_x=X, train_y=y, noise=1e-6, kernel="matern", optimizer="scipy", lengthscale_prior="DSLP")
    result = gp.fit(maxiter=200, x0=None)
    assert result["mll"] is not None and np.isfinite(result["mll"])
    assert result["params"].shape == (gp.num_hyperparams,)


def test_gp_predictions():
    """tests/test_gp.py:92-141."""
    X, y = quadratic_data(25, 2)
    gp = _GP(train_x=X, train_y=y, noise=1e-6)
    mean_single = gp.predict_mean_single(np.array([0.5, 0.5]))
    var_single = gp.predict_var_single(np.array([0.5, 0.5]))
    assert np.shape(mean_single) == () and np.shape(var_single) == ()
    assert var_single > 0
    pts = np.array([[0.2, 0.3], [0.7, 0.8], [0.5, 0.5]])
    means, variances = gp.predict_mean_batched(pts), gp.predict_var_batched(pts)
    assert means.shape == (3,) and variances.shape == (3,)
    assert np.all(variances > 0)
    var_at_train = gp.predict_var_single(X[0])
    assert var_at_train < 1e-3                                  # test_gp.py:139
    assert abs(gp.predict_mean_single(X[0]) - y[0, 0]) < 1e-2


def test_gp_update():
    """tests/test_gp.py:144-171: two new points are appended, a duplicate is rejected."""
    X, y = quadratic_data(15, 2)
    gp = _GP(train_x=X, train_y=y, noise=1e-6)
    n0 = gp.npoints
    new_X = np.array([[0.8, 0.2], [0.3, 0.9]])
    new_y = -np.sum((new_X - 0.5) ** 2, axis=1, keepdims=True)
    gp.update(new_X, new_y)
    assert gp.npoints == n0 + 2
    gp.update(new_X[0:1], new_y[0:1])
    assert gp.npoints == n0 + 2                                 # test_gp.py:169


def test_gp_random_point():
    """tests/test_gp.py:174-199."""
    X, y = quadratic_data(20, 3)
    gp = _GP(train_x=X, train_y=y)
    rng = np.random.default_rng(42)
    points = np.array([gp.get_random_point(rng=rng) for _ in range(10)])
    assert points.shape == (10, 3)
    assert np.all(points >= 0) and np.all(points <= 1)
    assert not np.allclose(points, points[0])


def test_gp_state_dict():
    """tests/test_gp.py:202-246."""
    from bobe_amd import GP
    X, y = quadratic_data(20, 2)
    gp1 = _GP(train_x=X, train_y=y, noise=1e-6, kernel="rbf", lengthscales=np.array([0.5, 0.3]), kernel_variance=2.0)
    gp2 = GP.from_state_dict(gp1.state_dict())
    assert gp2.ndim == gp1.ndim and gp2.npoints == gp1.npoints
    assert np.allclose(gp2.lengthscales, gp1.lengthscales)
    assert np.isclose(gp2.kernel_variance, gp1.kernel_variance)
    assert np.allclose(gp2.train_x, gp1.train_x)
    p = np.array([0.5, 0.5])
    assert np.isclose(gp1.predict_mean_single(p), gp2.predict_mean_single(p), rtol=1e-6)     # test_gp.py:244


def test_gp_copy():
    """tests/test_gp.py:249-272: a copy is independent of its source."""
    X, y = quadratic_data(15, 2)
    gp1 = _GP(train_x=X, train_y=y, noise=1e-6)
    gp2 = gp1.copy()
    gp2.update(np.array([[0.9, 0.1]]), np.array([[-0.5]]))
    assert gp1.npoints != gp2.npoints and gp2.npoints == gp1.npoints + 1


def test_gp_different_kernels():
    """tests/test_gp.py:275-295."""
    X, y = quadratic_data(20, 2)
    p = np.array([0.5, 0.5])
    m_rbf = _GP(train_x=X, train_y=y, kernel="rbf").predict_mean_single(p)
    m_mat = _GP(train_x=X, train_y=y, kernel="matern").predict_mean_single(p)
    assert not np.isclose(m_rbf, m_mat, rtol=0.01)


# --------------------------------------------------------------------------------- tests/test_acquisition.py
def test_ei_and_logei_initialization():
    """tests/test_acquisition.py:40-67."""
    from bobe_amd.acquisition import EI, LogEI
    ei, logei = EI(optimizer="scipy"), LogEI(optimizer="scipy")
    assert ei.name == "EI" and ei.optimizer == "scipy"
    assert logei.name == "LogEI" and logei.optimizer == "scipy"


def test_ei_and_logei_evaluation():
    """tests/test_acquisition.py:70-125: EI >= 0 everywhere; LogEI finite and consistent with EI."""
    from bobe_amd.acquisition import EI, LogEI
    gp = acquisition_gp(25, 2)
    ei, logei = EI(), LogEI()
    best_y = float(np.max(gp.train_y))
    for pt in (np.array([0.7, 0.7]), np.array([0.1, 0.1]), np.array([0.5, 0.5])):
        ei_val = -ei.fun(pt, gp, best_y, 0.0)
        assert ei_val >= 0                                       # test_acquisition.py:95
        logei_val = -logei.fun(pt, gp, best_y, 0.0)
        assert np.isfinite(logei_val)
        if ei_val > 1e-300:
            assert logei_val == pytest.approx(np.log(ei_val), rel=1e-6, abs=1e-6)


@pytest.mark.parametrize("which,seed", [("EI", 123), ("LogEI", 456)])
def test_ei_and_logei_optimization(which, seed):
    """tests/test_acquisition.py:128-201."""
    from bobe_amd import acquisition
    gp = acquisition_gp(20, 2, seed=seed)
    acq = getattr(acquisition, which)(optimizer="scipy")
    kw = {"best_y": float(np.max(gp.train_y)), "zeta": 0.0}
    pt, val = acq.get_next_point(gp=gp, acq_kwargs=kw, maxiter=100, n_restarts=5, verbose=False,
                                 rng=np.random.default_rng(42))
    assert pt.shape == (2,)
    assert np.all(pt >= 0) and np.all(pt <= 1)
    if which == "EI":
        assert val >= 0
    assert np.isfinite(gp.predict_mean_single(pt))


def test_batch_acquisition():
    """tests/test_acquisition.py:204-240: kriging-believer batch of three."""
    from bobe_amd.acquisition import EI
    gp = acquisition_gp(25, 2, seed=789)
    kw = {"best_y": float(np.max(gp.train_y)), "zeta": 0.0}
    pts, vals = EI(optimizer="scipy").get_next_batch(gp=gp, n_batch=3, acq_kwargs=kw, maxiter=100, n_restarts=3,
                                                     verbose=False, rng=np.random.default_rng(42))
    assert pts.shape == (3, 2) and vals.shape == (3,)
    assert np.all(pts >= 0) and np.all(pts <= 1)
    assert gp.npoints == 25                                      # the believer runs on a private copy


def test_acquisition_with_different_gp_settings():
    """tests/test_acquisition.py:243-272."""
    from bobe_amd.acquisition import EI
    gp_rbf = acquisition_gp(20, 2, seed=111)
    X, y = quadratic_data(20, 2, seed=222, centre=0.7)
    gp_matern = _GP(train_x=X, train_y=y, kernel="matern", noise=1e-6)
    ei = EI()
    best_y = float(np.max(gp_rbf.train_y))
    p = np.array([0.6, 0.6])
    assert -ei.fun(p, gp_rbf, best_y, 0.0) > 0 and -ei.fun(p, gp_matern, best_y, 0.0) > 0


def test_acquisition_optimization_convergence():
    """tests/test_acquisition.py:275-315: proposals stay near the optimum at (0.8, 0.8)."""
    from bobe_amd.acquisition import EI
    X, y = quadratic_data(30, 2, seed=333, centre=0.8)
    gp = _GP(train_x=X, train_y=y, noise=1e-6)
    ei = EI(optimizer="scipy")
    kw = {"best_y": float(np.max(gp.train_y)), "zeta": 0.0}
    rng = np.random.default_rng(42)
    pts = np.array([ei.get_next_point(gp=gp, acq_kwargs=kw, maxiter=100, n_restarts=5, verbose=False, rng=rng)[0]
                    for _ in range(3)])
    assert np.mean(np.linalg.norm(pts - np.array([0.8, 0.8]), axis=1)) < 0.5


# ------------------------------------------------------------------------------------------ tests/test_mpi.py
def test_gp_fit_serial():
    """tests/test_mpi.py:112-143: the pool's serial fit (one restart from the current hyper-parameters)."""
    from bobe_amd.bo import gp_fit
    gp = acquisition_gp(25, 2)
    result = gp_fit(gp, maxiters=200, n_restarts=1)
    assert np.isfinite(result["mll"])
    assert gp.lengthscales is not None and gp.kernel_variance > 0


def test_gp_state_serialization_for_pool():
    """tests/test_mpi.py:241-276: the state a worker receives rebuilds the same GP."""
    from bobe_amd import GP
    gp = acquisition_gp(20, 2)
    re = GP.from_state_dict(gp.state_dict())
    assert re.ndim == gp.ndim and re.npoints == gp.npoints
    assert np.allclose(re.lengthscales, gp.lengthscales)
    p = np.array([0.5, 0.5])
    assert np.isclose(gp.predict_mean_single(p), re.predict_mean_single(p), rtol=1e-6)


# ---------------------------------------------------------------------------------------- tests/test_clf_gp.py
def clf_data(n_good=30, n_bad=20, d=2, seed=42):
    """tests/test_clf_gp.py:18-37: a good region around 0.5 and a bad one in the corners, 10 log-units lower."""
    rng = np.random.RandomState(seed)
    Xg = rng.uniform(0.3, 0.7, size=(n_good, d))
    yg = -np.sum((Xg - 0.5) ** 2, axis=1, keepdims=True)
    Xb = rng.uniform(0, 1, size=(n_bad, d))
    Xb = np.where(Xb < 0.5, Xb * 0.4, 0.6 + Xb * 0.4)
    yb = -10 - np.sum((Xb - 0.5) ** 2, axis=1, keepdims=True)
    X, y = np.vstack([Xg, Xb]), np.vstack([yg, yb])
    perm = rng.permutation(len(y))
    return X[perm], y[perm]


def _CLF(*a, **k):
    from bobe_amd.clf_gp import GPwithClassifier
    return GPwithClassifier(*a, **k)


def test_clf_gp_initialization_svm():
    """tests/test_clf_gp.py:41-70 (the NN and ellipsoid classifiers of :73-130 are out of scope)."""
    X, y = clf_data(40, 30, 3)
    g = _CLF(train_x=X, train_y=y, clf_type="svm", clf_settings={"gamma": "scale", "C": 1e5}, clf_use_size=50,
             clf_threshold=5.0, gp_threshold=10.0, noise=1e-6)
    assert g.clf_type == "svm"
    assert g.clf_data_size == len(X)
    assert g.npoints <= len(X)
    assert g.use_clf
    with pytest.raises(ValueError):
        _CLF(train_x=X, train_y=y, clf_type="nn")


def test_clf_gp_predictions():
    """tests/test_clf_gp.py:133-188."""
    X, y = clf_data(40, 30, 2)
    g = _CLF(train_x=X, train_y=y, clf_type="svm", clf_use_size=50, clf_threshold=5.0, gp_threshold=10.0,
             probability_threshold=0.5, minus_inf=-1e5, noise=1e-6)
    good, bad = np.array([0.5, 0.5]), np.array([0.05, 0.05])
    mean_good, mean_bad = g.predict_mean_single(good), g.predict_mean_single(bad)
    assert g.predict_var_single(good) > 0 and g.predict_var_single(bad) > 0
    if g.use_clf:
        assert mean_bad < mean_good                             # test_clf_gp.py:170
    pts = np.array([[0.5, 0.5], [0.05, 0.05], [0.6, 0.4]])
    assert g.predict_mean_batched(pts).shape == (3,) and g.predict_var_batched(pts).shape == (3,)


def test_clf_gp_update():
    """tests/test_clf_gp.py:191-226."""
    X, y = clf_data(25, 15, 2)
    g = _CLF(train_x=X, train_y=y, clf_type="svm", clf_use_size=30, clf_threshold=5.0, gp_threshold=10.0, noise=1e-6)
    n_clf, n_gp = g.clf_data_size, g.npoints
    new_X = np.array([[0.55, 0.45], [0.48, 0.52]])
    g.update(new_X, -np.sum((new_X - 0.5) ** 2, axis=1, keepdims=True))
    assert g.clf_data_size == n_clf + 2
    assert g.npoints >= n_gp


def test_clf_gp_classifier_training():
    """tests/test_clf_gp.py:229-263: no classifier below clf_use_size, one after enough data arrived."""
    Xs, ys = clf_data(15, 10, 2)
    g = _CLF(train_x=Xs, train_y=ys, clf_type="svm", clf_use_size=50, clf_threshold=5.0, noise=1e-6)
    assert not g.use_clf
    Xa, ya = clf_data(25, 15, 2, seed=43)
    g.update(Xa, ya)
    g.train_classifier()
    if g.clf_data_size >= g.clf_use_size:
        assert g.use_clf


def test_clf_gp_random_point():
    """tests/test_clf_gp.py:266-296."""
    X, y = clf_data(40, 30, 2)
    g = _CLF(train_x=X, train_y=y, clf_type="svm", clf_use_size=50, clf_threshold=5.0, noise=1e-6)
    rng = np.random.default_rng(42)
    pts = np.array([g.get_random_point(rng=rng) for _ in range(5)])
    assert pts.shape == (5, 2) and np.all(pts >= 0) and np.all(pts <= 1)


def test_clf_gp_state_dict_and_copy():
    """tests/test_clf_gp.py:299-345 and :348-376."""
    from bobe_amd.clf_gp import GPwithClassifier
    X, y = clf_data(35, 25, 2)
    g1 = _CLF(train_x=X, train_y=y, clf_type="svm", clf_settings={"gamma": "scale", "C": 1e5}, clf_use_size=50,
              clf_threshold=5.0, gp_threshold=10.0, noise=1e-6)
    g2 = GPwithClassifier.from_state_dict(g1.state_dict())
    assert g2.clf_type == g1.clf_type and g2.clf_data_size == g1.clf_data_size and g2.use_clf == g1.use_clf
    assert np.allclose(g2.train_x_clf, g1.train_x_clf)
    p = np.array([0.5, 0.5])
    assert np.isclose(g1.predict_mean_single(p), g2.predict_mean_single(p), rtol=1e-5)
    g3 = g1.copy()
    g3.update(np.array([[0.6, 0.4]]), np.array([[-0.02]]))
    assert g1.clf_data_size != g3.clf_data_size and g3.clf_data_size == g1.clf_data_size + 1


# ------------------------------------------------------------------------------------------ tests/test_bo_2d.py
def _rosenbrock(x):
    return -((1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2)          # test_bo_2d.py:16-21


def _himmelblau(x):
    return -((x[0] ** 2 + x[1] - 11) ** 2 + (x[0] + x[1] ** 2 - 7) ** 2)  # test_bo_2d.py:24-29


_RESULT_KEYS = ("gp", "likelihood", "results_manager", "best_val", "best_pt", "termination_reason", "samples", "logz")


def test_bobe_ei_2d():
    """tests/test_bo_2d.py:32-100: the reference's constructor and run keywords, its result keys, EI on Rosenbrock."""
    from bobe_amd.bo import BOBE
    bobe = BOBE(loglikelihood=_rosenbrock, param_list=["x", "y"], param_bounds=np.array([[-2, 2], [-2, 2]]).T,
                likelihood_name="rosenbrock_ei_test", n_sobol_init=4, save=False, use_clf=False, seed=42,
                verbosity="WARNING")
    results = bobe.run(acq="ei", min_evals=15, max_evals=40, max_gp_size=40, ei_goal=1e-6, fit_n_points=5,
                       batch_size=1)
    for key in _RESULT_KEYS:
        assert key in results, key
    assert results["samples"] == {} and results["logz"] == {}           # test_bo_2d.py:82-83
    assert results["gp"].train_x.shape[0] >= 10 and results["gp"].train_x.shape[1] == 2
    assert results["best_val"] > -1000
    assert results["best_pt"].shape == (2,)
    assert isinstance(results["termination_reason"], str)


def test_bobe_wipstd_2d():
    """tests/test_bo_2d.py:103-192: WIPStd on Himmelblau with the logZ convergence test on the surrogate."""
    from bobe_amd.bo import BOBE
    bobe = BOBE(loglikelihood=_himmelblau, param_list=["x", "y"], param_bounds=np.array([[-5, 5], [-5, 5]]).T,
                likelihood_name="himmelblau_test", n_sobol_init=4, save=False, use_clf=False, seed=123,
                verbosity="WARNING")
    results = bobe.run(acq="wipstd", min_evals=25, max_evals=60, max_gp_size=60, logz_threshold=0.5,
                       convergence_n_iters=2, fit_n_points=8, ns_n_points=15, batch_size=1,
                       mc_points_method="uniform")
    for key in _RESULT_KEYS:
        assert key in results, key
    assert results["gp"].train_x.shape[0] >= 30 and results["gp"].train_x.shape[1] == 2
    assert results["best_val"] > -500
    if results["samples"]:
        assert len(results["samples"]["x"]) > 0
    assert {"mean", "upper", "lower"} <= set(results["logz"])


def test_bobe_with_classifier(tmp_path):
    """tests/test_bo_2d.py:195-243, plus the save keyword: the GP checkpoint is written and reloads."""
    from bobe_amd.bo import BOBE
    from bobe_amd.clf_gp import GPwithClassifier
    bobe = BOBE(loglikelihood=_rosenbrock, param_list=["x", "y"], param_bounds=np.array([[-2, 2], [-2, 2]]).T,
                likelihood_name="rosenbrock_clf_test", n_sobol_init=4, save=True, save_dir=str(tmp_path),
                use_clf=True, clf_type="svm", clf_use_size=10, seed=456, verbosity="WARNING")
    results = bobe.run(acq="wipstd", min_evals=20, max_evals=50, max_gp_size=50, logz_threshold=0.5, fit_n_points=6,
                       ns_n_points=12, batch_size=1)
    assert "gp" in results and hasattr(results["gp"], "use_clf")
    assert results["best_pt"].shape == (2,) and np.isfinite(results["best_val"])
    re = GPwithClassifier.load(str(tmp_path / "rosenbrock_clf_test_gp"))
    assert re.clf_data_size == results["gp"].clf_data_size
