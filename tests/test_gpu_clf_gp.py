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
    # the probability function rebuilt from the stored parameters (clf.py:71-78, the load path) is the trained one
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


def _svm_case(d, n, seed, width):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = -2000.0 * np.sum((X - 0.5) ** 2, axis=1, keepdims=True)
    return rng, X, y, width


@pytest.mark.parametrize("d,n,seed,width", [(2, 120, 0, 150.0), (6, 400, 1, 450.0), (10, 900, 2, 750.0), (20, 300, 3, 2500.0)])
def test_device_decision_function_against_sklearn_and_the_direct_difference_restatement(d, n, seed, width):
    """The gate's decision values (bobe_gp_gate_eval, k_gate: direct differences in one fixed summation order) against
    (i) scikit-learn's own ``SVC.decision_function`` (libsvm: the |x|^2 + |sv|^2 - 2 x.sv expansion) and (ii) the oracle's
    line-for-line restatement of clf.py:188-213.  With C = 1e7 the terms dual_i k_i are large and cancel, so two
    summation orders differ by a few ulp of sum_i |dual_i| k_i ("scale"), not of the decision: the stated tolerance is
    |delta| <= 1e-9 |decision| + 1e-13 scale; no decision may change sign unless it lies inside that band."""
    from sklearn.svm import SVC
    from bobe_amd.clf_gp import GPwithClassifier
    from oracle import bobe_oracle_loop as OL
    rng, X, y, width = _svm_case(d, n, seed, width)
    g = GPwithClassifier(X, y, clf_threshold=width, gp_threshold=2 * width, noise=1e-6, lengthscales=np.full(d, 0.4))
    assert g.use_clf and g.clf_params is not None
    labels = np.where(y.ravel() < y.max() - width, 0, 1)
    ref = SVC(kernel="rbf", gamma="scale", C=1e7).fit(X, labels)
    p = g.clf_params
    assert np.array_equal(p["support_vectors"], ref.support_vectors_) and np.array_equal(p["dual_coef"], ref.dual_coef_[0])
    q = np.vstack([rng.uniform(size=(3000, d)), X[:50], ref.support_vectors_[:20]])
    dev = g.clf_decision(q)
    sk = ref.decision_function(q)
    orc = OL.svm_predict(q, p["support_vectors"], p["dual_coef"], p["intercept"], p["gamma_eff"])
    scale = OL.svm_decision_scale(q, p["support_vectors"], p["dual_coef"], p["gamma_eff"])
    for name, other in (("sklearn", sk), ("oracle", orc)):
        tol = 1e-9 * np.abs(other) + 1e-13 * scale
        err = np.abs(dev - other)
        assert np.all(err <= tol), (name, float(np.max(err / tol)))
        flips = (dev >= 0) != (other >= 0)
        assert not np.any(flips & (np.abs(other) > tol)), name
    # the probabilities (clf.py:210-213) and the gate itself against the ORACLE's own values, not the product's
    probs = OL.svm_predict_proba(q, p["support_vectors"], p["dual_coef"], p["intercept"], p["gamma_eff"])
    clear = np.abs(orc) > 1e-9 * np.abs(orc) + 1e-13 * scale
    assert np.array_equal(g._clf_predict_func(q)[clear], probs[clear])
    assert 0 < probs.sum() < len(probs)


def test_gate_inside_every_entry_point_against_the_oracle_gate():
    """clf_gp.py:173-205 with the ORACLE's probabilities (svm_predict_proba restated from clf.py:188-213) and the oracle's
    GP: predict_mean / predict_var (physical), predict_single (standardised), the posterior gradients (zero where
    gated) and EI / LogEI (the gated predict_single inside EI.fun, acquisition.py:246, 323)."""
    from bobe_amd.clf_gp import GPwithClassifier
    from oracle import bobe_oracle as O
    from oracle import bobe_oracle_loop as OL
    rng = np.random.default_rng(11)
    d = 3
    X = rng.uniform(size=(150, d))
    y = -600.0 * np.sum((X - 0.5) ** 2, axis=1)
    gp = GPwithClassifier(X, y, clf_threshold=80.0, gp_threshold=160.0, noise=1e-6, lengthscales=np.full(d, 0.4),
                          minus_inf=-1e10)
    assert gp.use_clf
    mask = y > y.max() - 160.0
    og = O.OracleGP(X[mask], y[mask], noise=1e-6, lengthscales=np.full(d, 0.4), lengthscale_prior="DSLP")
    q = rng.uniform(size=(400, d))
    p = gp.clf_params
    probs = OL.svm_predict_proba(q, p["support_vectors"], p["dual_coef"], p["intercept"], p["gamma_eff"])
    ok = probs >= 0.5
    assert 20 < ok.sum() < len(q) - 20
    wm, wv = OL.clf_gate(og.predict_mean_batched(q), og.predict_var_batched(q), probs, 0.5, -1e10)
    gm, gv = gp.predict_mean_batched(q), gp.predict_var_batched(q)
    assert np.array_equal(gm[~ok], wm[~ok]) and np.array_equal(gv[~ok], wv[~ok])          # exactly minus_inf / 1e-12
    assert np.allclose(gm[ok], wm[ok], rtol=1e-7, atol=1e-6) and np.allclose(gv[ok], wv[ok], rtol=1e-6, atol=1e-9 * og.y_std ** 2)
    ms, vs = og.predict_batched(q)
    wm2, wv2 = OL.clf_gate(ms, vs, probs, 0.5, -1e10)
    m, v = gp.predict_batched(q)
    assert np.array_equal(m[~ok], wm2[~ok]) and np.array_equal(v[~ok], wv2[~ok])
    assert np.allclose(m[ok], wm2[ok], rtol=1e-7, atol=1e-7) and np.allclose(v[ok], wv2[ok], rtol=1e-6, atol=1e-12)
    # gradients: zero where gated, the plain GP's elsewhere
    from bobe_amd import GP
    plain = GP(X[mask], y[mask], noise=1e-6, lengthscales=np.full(d, 0.4), lengthscale_prior="DSLP")
    for mean_only in (True, False):
        a = gp.predict_grad(q, mean_only=mean_only)
        b = plain.predict_grad(q, mean_only=mean_only)
        assert np.array_equal(a[0][~ok], np.full((~ok).sum(), -1e10)) and np.array_equal(a[0][ok], b[0][ok])
        assert np.all(a[2][~ok] == 0.0) and np.array_equal(a[2][ok], b[2][ok])
        if not mean_only:
            assert np.all(a[1][~ok] == 1e-12) and np.all(a[3][~ok] == 0.0) and np.array_equal(a[3][ok], b[3][ok])
    # EI / LogEI from the gated predict_single
    best = float(np.max(gp.train_y))
    for log_ei in (False, True):
        got = gp.acq_ei(q, best, 0.01, log_ei=log_ei)
        want = O.log_ei_score(wm2, wv2, best, 0.01) if log_ei else O.ei_score(wm2, wv2, best, 0.01)
        assert np.allclose(got[ok], want[ok], rtol=1e-6, atol=1e-12)
        assert np.allclose(got[~ok], want[~ok], rtol=1e-9, atol=0.0) if log_ei else np.all(got[~ok] == 0.0)
    # switching the classifier off clears the gate in the library; back on restores it
    gp.use_clf = False
    assert np.allclose(gp.predict_mean_batched(q), plain.predict_mean_batched(q), atol=1e-9)
    gp.use_clf = True
    assert np.array_equal(gp.predict_mean_batched(q)[~ok], wm[~ok])


def test_gated_hmc_runs_on_the_device_and_matches_the_host_stepped_path():
    """A gated GP now keeps the device sampler (bobe_gp_hmc_run / _leapfrog with the gate inside the kernels).  (i) One
    fused trajectory equals the same trajectory stepped from the host through bobe_gp_predict_grad (whose gate is the same
    device function): gated end points carry mean = minus_inf and are never accepted.  (ii) The samples of the device
    chains stay inside the feasible region and their moments agree with the host-stepped sampler's."""
    from scipy.special import expit
    from bobe_amd import samplers
    from bobe_amd.clf_gp import GPwithClassifier
    rng = np.random.default_rng(5)
    d = 2
    X = rng.uniform(size=(200, d))
    y = -800.0 * np.sum((X - np.array([0.45, 0.55])) ** 2, axis=1)
    gp = GPwithClassifier(X, y, clf_threshold=40.0, gp_threshold=120.0, noise=1e-6, lengthscales=np.full(d, 0.3),
                          minus_inf=-1e10)
    assert gp.use_clf
    # (i) one trajectory, L steps, from points on both sides of the gate
    P, L, eps = 64, 6, 0.15
    x0 = rng.uniform(0.02, 0.98, size=(P, d))
    U = np.log(x0) - np.log1p(-x0)
    inv_mass = np.ones(d)

    def host_logp_grad(Uc):
        Xc = np.clip(expit(Uc), 1e-12, 1 - 1e-12)
        m, _, dm, _ = gp.predict_grad(Xc, mean_only=True)
        bad = m <= gp.minus_inf
        mean = np.where(bad, gp.minus_inf, m * gp.y_std + gp.y_mean)
        gx = np.where(bad[:, None], 0.0, dm * gp.y_std)
        return mean + np.sum(np.log(Xc) + np.log1p(-Xc), axis=1), gx * (Xc * (1 - Xc)) + (1 - 2 * Xc), mean, Xc

    _, g0, _, _ = host_logp_grad(U)
    p0 = rng.normal(size=U.shape)
    Un, pn, lpn, gn, meann, Xn = gp.hmc_leapfrog(U, p0 + 0.5 * eps * g0, inv_mass, eps, L, 1.0)
    Uh, ph = U.copy(), p0 + 0.5 * eps * g0
    for s_ in range(L):
        Uh = Uh + eps * inv_mass * ph
        lph, gh, meanh, Xh = host_logp_grad(Uh)
        ph = ph + (eps if s_ < L - 1 else 0.5 * eps) * gh
    gated_end = meanh <= gp.minus_inf
    assert 0 < gated_end.sum() < P
    assert np.array_equal(meann <= gp.minus_inf, gated_end) and np.all(meann[gated_end] == gp.minus_inf)
    assert np.allclose(Un, Uh, rtol=1e-9, atol=1e-9) and np.allclose(lpn[~gated_end], lph[~gated_end], rtol=1e-9, atol=1e-7)
    assert np.allclose(gn, gh, rtol=1e-8, atol=1e-8)
    # (ii) whole chains on the device vs the host-stepped sampler
    dev = samplers.sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), num_chains=4, warmup_steps=200, num_samples=800, thinning=2)
    host = samplers.sample_GP_NUTS(gp, np_rng=np.random.default_rng(2), num_chains=4, warmup_steps=200, num_samples=800, thinning=2,
                                   fused_trajectories=False)
    for smp in (dev, host):
        assert np.all(gp._clf_predict_func(smp["x"]) >= 0.5)            # no sample outside the classifier's region
        assert np.all(smp["logp"] > gp.minus_inf)
    assert np.allclose(dev["x"].mean(0), host["x"].mean(0), atol=0.02)
    assert np.allclose(dev["x"].std(0), host["x"].std(0), rtol=0.25)


def test_gate_entry_points_through_the_c_abi():
    """bobe_gp_set_gate / bobe_gp_gate_eval as a host program would call them: host or device pointers, NULL clears,
    usage errors come back as status codes with text, the probability threshold is honoured, a NaN query is infeasible,
    and the gate survives set_data / factor but is not copied by bobe_gp_clone_state."""
    import ctypes as C
    import torch
    from bobe_amd import GP, _lib
    rng = np.random.default_rng(4)
    d, n_sv = 5, 37
    sv = np.ascontiguousarray(rng.uniform(size=(n_sv, d)))
    dual = np.ascontiguousarray(rng.normal(size=n_sv) * 50.0)
    b, gamma = 0.3, 2.5
    X = rng.uniform(size=(60, d))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-6, lengthscales=np.full(d, 0.5))
    lib, h = gp._lib, gp._h
    q = np.ascontiguousarray(rng.uniform(size=(300, d)))
    dec, ok = np.empty(300), np.empty(300)
    # no gate yet: an error code and a message, nothing thrown
    rc = lib.bobe_gp_gate_eval(h, _lib.ptr(q), 300, _lib.ptr(dec), _lib.ptr(ok))
    assert rc < 0 and b"no classifier gate" in lib.bobe_last_error()
    assert lib.bobe_gp_set_gate(h, _lib.ptr(sv), n_sv, None, b, gamma, 0.5, -1e5) < 0          # dual_coef missing
    assert lib.bobe_gp_set_gate(h, _lib.ptr(sv), n_sv, _lib.ptr(dual), b, gamma, 0.5, -1e5) == 0
    assert lib.bobe_gp_gate_eval(h, _lib.ptr(q), 300, _lib.ptr(dec), _lib.ptr(ok)) == 0
    want = np.exp(-gamma * ((q[:, None, :] - sv[None, :, :]) ** 2).sum(-1)) @ dual + b
    scale = np.exp(-gamma * ((q[:, None, :] - sv[None, :, :]) ** 2).sum(-1)) @ np.abs(dual)
    assert np.all(np.abs(dec - want) <= 1e-9 * np.abs(want) + 1e-13 * scale)
    assert np.array_equal(ok, (dec >= 0).astype(float)) and 0 < ok.sum() < 300
    # device pointers in and out (a torch tensor's memory), either output optional
    qd = torch.from_numpy(q).cuda()
    dd = torch.empty(300, dtype=torch.float64, device="cuda")
    assert lib.bobe_gp_gate_eval(h, C.c_void_p(qd.data_ptr()), 300, C.c_void_p(dd.data_ptr()), None) == 0
    assert np.array_equal(dd.cpu().numpy(), dec)
    # the gated predictions: -inf marks / 1e-12, untouched elsewhere; NaN coordinates are infeasible
    m, v = gp.predict_batched(q)
    assert np.all(np.isneginf(m[ok == 0])) and np.all(v[ok == 0] == 1e-12) and np.all(np.isfinite(m[ok == 1]))
    qn = q[:4].copy()
    qn[1, 2] = np.nan
    assert lib.bobe_gp_gate_eval(h, _lib.ptr(qn), 4, _lib.ptr(dec[:4].copy()), _lib.ptr(ok[:4])) == 0 and ok[1] == 0.0
    # probability threshold (clf_gp.py:179): proba is 0 or 1, so <= 0 lets everything through and > 1 nothing
    for thr, expect in ((0.0, 300), (1.5, 0)):
        assert lib.bobe_gp_set_gate(h, _lib.ptr(sv), n_sv, _lib.ptr(dual), b, gamma, thr, -1e5) == 0
        assert lib.bobe_gp_gate_eval(h, _lib.ptr(q), 300, None, _lib.ptr(ok)) == 0 and int(ok.sum()) == expect
    assert lib.bobe_gp_set_gate(h, _lib.ptr(sv), n_sv, _lib.ptr(dual), b, gamma, 0.5, -1e5) == 0
    # new data / a new factor keep the gate; a clone does not carry it
    gp.update(rng.uniform(size=(3, d)), rng.normal(size=(3, 1)))
    m2, _ = gp.predict_batched(q)
    assert np.array_equal(np.isneginf(m2), np.isneginf(m))
    twin = gp.copy()
    assert np.all(np.isfinite(twin.predict_batched(q)[0]))
    # NULL clears
    assert lib.bobe_gp_set_gate(h, None, 0, None, 0.0, 0.0, 0.5, -1e5) == 0
    assert np.all(np.isfinite(gp.predict_batched(q)[0]))


def test_clf_module_functions_of_the_reference():
    """bobe_amd.clf mirrors the SVM half of BOBE/clf.py: ``train_svm_classifier`` returns the reference's triple (params,
    metrics, predict_proba_fn), ``svm_predict`` / ``svm_predict_proba`` (clf.py:188-213) evaluate the decision function on
    the device - against scikit-learn's own decision_function and the oracle's line-for-line restatement."""
    from sklearn.svm import SVC
    from bobe_amd import clf
    from oracle import bobe_oracle_loop as OL
    rng = np.random.default_rng(2)
    X = rng.uniform(size=(200, 3))
    Y = (np.sum((X - 0.5) ** 2, axis=1) < 0.2).astype(int)
    params, metrics, proba_fn = clf.train_svm_classifier(X, Y, settings={})
    assert set(params) >= {"support_vectors", "dual_coef", "intercept", "gamma_eff"} and metrics["n_support_vectors"] > 0
    assert set(clf.CLASSIFIER_REGISTRY) == {"svm"} and clf.CLASSIFIER_REGISTRY["svm"]["train_fn"] is clf.train_svm_classifier
    q = rng.uniform(size=(64, 3))
    sk = SVC(kernel="rbf", gamma="scale", C=1e7).fit(X, Y)
    dec = clf.svm_predict(q, params["support_vectors"], params["dual_coef"], params["intercept"], params["gamma_eff"])
    ref = sk.decision_function(q)
    mag = np.array([np.sum(np.abs(params["dual_coef"]) * np.exp(-params["gamma_eff"] * np.sum((params["support_vectors"] - p) ** 2, 1)))
                    for p in q])
    assert np.all(np.abs(dec - ref) <= 1e-9 * np.abs(ref) + 1e-13 * mag)
    one = clf.svm_predict(q[0], params["support_vectors"], params["dual_coef"], params["intercept"], params["gamma_eff"])
    assert isinstance(one, float) and one == dec[0]
    want = OL.svm_predict(q[0], params["support_vectors"], params["dual_coef"], params["intercept"], params["gamma_eff"])
    assert abs(one - want) <= 1e-9 * abs(want) + 1e-13 * mag[0]
    pr = clf.svm_predict_proba(q, params["support_vectors"], params["dual_coef"], params["intercept"], params["gamma_eff"])
    clear = np.abs(ref) > 1e-9 * mag                                  # (not within rounding of the boundary)
    assert np.array_equal(pr[clear], (ref[clear] >= 0).astype(float))
    assert np.array_equal(proba_fn(q), pr) and np.array_equal(clf.get_svm_predict_proba_fn(params)(q), pr)
