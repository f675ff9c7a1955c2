"""Callers either side of the hot path on the GPU against the oracle's twins (rows a19/a20):
every step of the frozen oracle BO run (tests/golden/loop_himmelblau.npz) replayed from its recorded state; the logZ
dictionary of a finished nested-sampling run on fixed dead points; the classifier gate."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loop_himmelblau.npz")


def _gpu_gp(g, it):
    from bobe_amd import GP
    return GP(g[f"s{it}_train_x"], g[f"s{it}_train_y"], lengthscales=g[f"s{it}_lengthscales"],
              kernel_variance=float(g[f"s{it}_kernel_variance"]))


@pytest.mark.parametrize("it", range(5))
def test_believer_batch_replays_the_oracle_step(it):
    """get_mc_points -> sweep -> argmin -> local refinement -> believer update, member by member
    (acquisition.py:147-196, 350-412): same integration points, same sweep pick, same refined point."""
    from bobe_amd.acquisition import WIPStd, get_mc_points
    g = np.load(GOLD)
    gp = _gpu_gp(g, it)
    acq = WIPStd()
    mc = {"x": g[f"s{it}_mc_samples"]}
    n_batch, msize = int(g["n_batch"][it]), int(g["mc_points_size"])
    # first member's sweep on its own: exact index (tie rule of SURVEY 7.4: a different index is accepted only if
    # the two scores agree to 1e-9 relative)
    pts = get_mc_points(mc, msize, rng=np.random.default_rng(1000 + it))
    assert np.array_equal(pts, g[f"s{it}_mc_points"][0])
    scores, idx = acq.sweep(gp, pts, pts)
    want_idx = int(g[f"s{it}_sweep_index"][0])
    assert idx == want_idx or abs(scores[idx] - scores[want_idx]) <= 1e-9 * abs(scores[want_idx])
    assert scores[want_idx] == pytest.approx(float(g[f"s{it}_sweep_value"][0]), rel=1e-7)
    xs, vals = acq.get_next_batch(gp, n_batch=n_batch, acq_kwargs={"mc_samples": mc, "mc_points_size": msize},
                                  n_restarts=1, maxiter=100, rng=np.random.default_rng(1000 + it))
    assert xs.shape == (n_batch, 2)
    # the oracle refines with central differences, the GPU with exact gradients: the two L-BFGS-B runs stop within the
    # optimiser's own tolerance of each other; the scores must agree much more tightly than the points
    assert np.allclose(xs, g[f"s{it}_batch_x"], atol=2e-3), (xs, g[f"s{it}_batch_x"])
    assert np.allclose(vals, g[f"s{it}_batch_val"], rtol=2e-5)


@pytest.mark.parametrize("it", range(5))
def test_update_and_refit_replay_the_oracle_step(it):
    """update_gp (bo.py:620-668): duplicate filter + re-standardisation + refactor, then the policy's multi-restart
    fit with the recorded generator -> the oracle's hyper-parameters."""
    from bobe_amd.bo import gp_fit, refit_policy
    g = np.load(GOLD)
    gp = _gpu_gp(g, it)
    refit, n_restarts, maxiter, n_since = refit_policy(gp.train_x.shape[0], int(g[f"s{it}_n_since"]),
                                                       g[f"s{it}_batch_x"].shape[0], int(g["fit_n_points"]))
    assert (refit, n_restarts, maxiter) == (bool(g[f"s{it}_refit"]), int(g[f"s{it}_n_restarts"]), int(g[f"s{it}_maxiter"]))
    gp.update(g[f"s{it}_batch_x"], g[f"s{it}_new_y"])
    assert gp.npoints == g[f"s{it}_train_x"].shape[0] + g[f"s{it}_batch_x"].shape[0]
    if refit:
        gp_fit(gp, maxiters=maxiter, n_restarts=n_restarts, rng=np.random.default_rng(2000 + it))
    f = gp.neg_mll(np.log(gp.get_hyperparams()))
    assert f == pytest.approx(float(g[f"s{it}_after_neg_mll"]), rel=1e-6, abs=1e-6)
    assert np.allclose(gp.lengthscales, g[f"s{it}_after_lengthscales"], rtol=2e-3)
    assert gp.kernel_variance == pytest.approx(float(g[f"s{it}_after_kernel_variance"]), rel=5e-3)


def test_logz_dictionary_against_the_oracle_twin():
    """samplers.py:172-183 on FIXED dead points and log-volumes: variance from the GPU GP vs from the OracleGP."""
    from bobe_amd import GP
    from bobe_amd.samplers import compute_integrals, logz_from_samples
    from oracle import bobe_oracle as O
    from oracle import bobe_oracle_loop as OL
    rng = np.random.default_rng(11)
    X = rng.uniform(size=(60, 3))
    y = -8.0 * np.sum((X - 0.4) ** 2, axis=1)
    ls = np.array([0.5, 0.7, 0.6])
    gp = GP(X, y, noise=1e-6, lengthscales=ls, kernel_variance=2.0)
    og = O.OracleGP(X, y, noise=1e-6, lengthscales=ls, kernel_variance=2.0)
    nlive, n = 50, 600
    sx = rng.uniform(size=(n, 3))
    order = np.argsort(og.predict_mean_batched(sx))
    sx = sx[order]
    logl = og.predict_mean_batched(sx)
    logvol = -np.arange(1, n + 1) / nlive
    mean = float(OL.compute_integrals(logl, logvol)[-1])
    assert compute_integrals(logl, logvol)[-1] == pytest.approx(mean, abs=1e-12)
    got = logz_from_samples(gp, sx, logl, logvol, mean, 0.0)
    want = OL.logz_bounds(logl, logvol, og.predict_var_batched(sx), mean)
    for k in ("upper", "lower"):
        assert got[k] == pytest.approx(want[k], abs=1e-8)
    assert got["var"] == pytest.approx(want["var"], rel=1e-6) and got["std"] == pytest.approx(want["std"], rel=1e-6)
    assert got["lower"] < got["mean"] < got["upper"]


def test_classifier_gate_against_the_oracle_twin():
    """clf_gp.py:86-93, 173-205: GP on the thresholded subset, predictions gated by the classifier's probability.
    The SVM is TRAINED by scikit-learn on both sides; its probabilities come from the oracle's restatement of
    clf.py:188-213 (not from the product), and the subset, the labels and the gate are compared."""
    from bobe_amd.clf_gp import GPwithClassifier
    from oracle import bobe_oracle as O
    from oracle import bobe_oracle_loop as OL
    rng = np.random.default_rng(3)
    X = rng.uniform(size=(80, 2))
    y = -400.0 * np.sum((X - 0.5) ** 2, axis=1)               # spans ~200: some points beyond clf_threshold = 60
    gp = GPwithClassifier(X, y, clf_threshold=60.0, gp_threshold=120.0, noise=1e-6, lengthscales=[0.4, 0.4],
                          minus_inf=-1e10)
    assert gp.use_clf
    mask = y > y.max() - 120.0
    assert gp.npoints == int(mask.sum())
    og = O.OracleGP(X[mask], y[mask], noise=1e-6, lengthscales=np.array([0.4, 0.4]), lengthscale_prior="DSLP")
    q = rng.uniform(size=(200, 2))
    p = gp.clf_params
    probs = OL.svm_predict_proba(q, p["support_vectors"], p["dual_coef"], p["intercept"], p["gamma_eff"])
    assert np.array_equal(gp._clf_predict_func(q), probs)                         # (no query of this set sits on the boundary)
    assert 0 < np.sum(probs >= 0.5) < len(q)                                      # both sides of the gate are exercised
    labels = OL.clf_labels(y, 60.0)
    assert np.array_equal(labels, np.where(y < y.max() - 60.0, 0, 1))
    wm, wv = OL.clf_gate(og.predict_mean_batched(q), og.predict_var_batched(q), probs, 0.5, -1e10)
    assert np.allclose(gp.predict_mean_batched(q), wm, rtol=1e-7, atol=1e-6)
    assert np.allclose(gp.predict_var_batched(q), wv, rtol=1e-6, atol=1e-9 * og.y_std ** 2)
    m, v = gp.predict_batched(q)
    ms, vs = og.predict_batched(q)
    wm2, wv2 = OL.clf_gate(ms, vs, probs, 0.5, -1e10)
    assert np.allclose(m, wm2, rtol=1e-7, atol=1e-7) and np.allclose(v, wv2, rtol=1e-6, atol=1e-12)


def test_classifier_is_retrained_as_the_best_value_moves():
    """bo.py:673-676: update_gp retrains the classifier after every update; labels follow train_y_clf.max()."""
    from bobe_amd.bo import BOBE

    def like(x):
        return -300.0 * float(np.sum((np.asarray(x) - 0.6) ** 2))

    b = BOBE(like, ["a", "b"], np.array([[0.0, 1.0], [0.0, 1.0]]).T, n_sobol_init=24, use_clf=True,
             clf_nsigma_threshold=5, seed=5, save=False)
    gp = b.gp
    thr = gp.clf_threshold
    before = np.where(gp.train_y_clf.flatten() < gp.train_y_clf.max() - thr, 0, 1).copy()
    # a much better point moves the maximum: points that were feasible fall out of the band
    b.update_gp(np.array([[0.6, 0.6]]), np.array([[like([0.6, 0.6]) + 150.0]]))
    after = np.where(gp.train_y_clf.flatten() < gp.train_y_clf.max() - thr, 0, 1)
    assert after[:-1].sum() < before.sum()
    if gp.use_clf:
        # the SVM in use was fitted to the NEW labels (train_classifier ran inside update_gp)
        pred = (gp._clf_predict_func(gp.train_x_clf) >= 0.5).astype(int)
        assert np.mean(pred == after) > 0.9
        assert gp.clf_metrics is not None
