"""Pins the CPU oracle (oracle/bobe_oracle.py) against independent implementations.

The reference (BOBE/gp.py, BOBE/acquisition.py) cannot be imported here and its tests hold no
numeric golden vectors, so the oracle is checked against scipy.linalg / scipy.stats /
scipy.special, torch fp64 autograd, finite differences, closed forms, and the reference tests'
own invariants (tests/test_gp.py, tests/test_acquisition.py data recipes).
"""
import math

import numpy as np
import pytest
import scipy.stats as st
from scipy.linalg import cho_factor, cho_solve

from oracle import bobe_oracle as O


def ref_test_data(n=50, d=2, seed=42):
    # reference tests/test_gp.py:21-27
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(n, d))
    y = -np.sum((X - 0.5) ** 2, axis=1).reshape(-1, 1)
    return X, y


def test_kernel_closed_forms():
    rng = np.random.default_rng(0)
    A, B = rng.uniform(size=(7, 3)), rng.uniform(size=(5, 3))
    ls = np.array([0.3, 0.7, 1.1])
    K = O.rbf_kernel(A, B, ls, 2.0, 1e-6, include_noise=False)
    for i in range(7):
        for j in range(5):
            r2 = np.sum(((A[i] - B[j]) / ls) ** 2)
            assert K[i, j] == pytest.approx(2.0 * math.exp(-0.5 * r2), rel=1e-14)
    Km = O.matern_kernel(A, B, ls, 2.0, 1e-6, include_noise=False)
    r = math.sqrt(np.sum(((A[2] - B[3]) / ls) ** 2))
    assert Km[2, 3] == pytest.approx(2.0 * (1 + math.sqrt(5) * r + 5 / 3 * r * r) * math.exp(-math.sqrt(5) * r), rel=1e-14)
    Kxx = O.rbf_kernel(A, A, ls, 2.0, 1e-3, include_noise=True)
    assert np.allclose(np.diag(Kxx), 2.0 + 1e-3, rtol=0, atol=0)        # exact zero distance on the diagonal
    Kmm = O.matern_kernel(A, A, ls, 2.0, 1e-3, include_noise=True)
    assert np.allclose(np.diag(Kmm), 2.0 * (1 + 1e-15 * math.sqrt(5)) * math.exp(-math.sqrt(5) * 1e-15) + 1e-3, rtol=1e-14)


def test_mll_against_scipy_and_closed_form():
    X, y = ref_test_data(30, 3)
    ys = (y - y.mean()) / y.std()
    ls = np.array([0.5, 0.8, 1.2])
    K = O.rbf_kernel(X, X, ls, 1.3, 1e-6)
    c = cho_factor(K, lower=True)
    alpha = cho_solve(c, ys)
    want = -0.5 * float((ys.T @ alpha)[0, 0]) - np.sum(np.log(np.diag(c[0]))) - 0.5 * 30 * math.log(2 * math.pi)
    assert O.gp_mll(K, ys, 30) == pytest.approx(want, rel=1e-13)
    # multivariate normal log-pdf is the same number
    assert O.gp_mll(K, ys, 30) == pytest.approx(st.multivariate_normal(np.zeros(30), K, allow_singular=True).logpdf(ys.ravel()), rel=1e-9)
    # N=1 closed form
    assert O.gp_mll(np.array([[2.0]]), np.array([[0.7]]), 1) == pytest.approx(
        -0.5 * 0.49 / 2.0 - 0.5 * math.log(2.0) - 0.5 * math.log(2 * math.pi), rel=1e-14)
    # not PD -> NaN, no exception
    assert math.isnan(O.gp_mll(np.array([[1.0, 2.0], [2.0, 1.0]]), np.ones((2, 1)), 2))


@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_mll_gradient_vs_torch_autograd_and_fd(kernel):
    import torch
    rng = np.random.default_rng(3)
    n, d = 60, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - X[:, 2]
    y = (y - y.mean()) / y.std()
    ls = np.array([0.4, 0.9, 0.6])
    kvar, noise = 1.7, 1e-6
    mll, g = O.mll_value_and_grad(kernel, X, y, ls, kvar, noise)

    th = torch.tensor(np.log(np.append(ls, kvar)), dtype=torch.float64, requires_grad=True)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    l_, kv = torch.exp(th[:d]), torch.exp(th[d])
    Xs = Xt / l_
    dsq = ((Xs[:, None, :] - Xs[None, :, :]) ** 2).sum(-1)
    if kernel == "rbf":
        K = kv * torch.exp(-0.5 * dsq)
    else:
        r = torch.sqrt(torch.where(dsq < 1e-30, torch.full_like(dsq, 1e-30), dsq))
        K = kv * (1 + r * (math.sqrt(5) + r * 5 / 3)) * torch.exp(-math.sqrt(5) * r)
    K = K + noise * torch.eye(n, dtype=torch.float64)
    L = torch.linalg.cholesky(K)
    a = torch.cholesky_solve(yt[:, None], L)
    val = -0.5 * (yt[None, :] @ a)[0, 0] - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2 * math.pi)
    val.backward()
    assert mll == pytest.approx(val.item(), rel=1e-12)
    assert np.max(np.abs(g - th.grad.numpy())) <= 1e-8 * np.max(np.abs(g))
    # central finite differences
    for i in range(d + 1):
        t = np.log(np.append(ls, kvar))
        tp, tm = t.copy(), t.copy()
        tp[i] += 1e-5
        tm[i] -= 1e-5
        fp = O.mll_value_and_grad(kernel, X, y, np.exp(tp[:d]), math.exp(tp[d]), noise)[0]
        fm = O.mll_value_and_grad(kernel, X, y, np.exp(tm[:d]), math.exp(tm[d]), noise)[0]
        assert g[i] == pytest.approx((fp - fm) / 2e-5, rel=2e-5, abs=1e-6)


def test_priors_against_scipy_stats():
    x = np.array([0.05, 0.7, 3.0])
    assert np.allclose(O._logpdf_lognormal(x, 0.3, 1.7), st.lognorm(s=1.7, scale=math.exp(0.3)).logpdf(x), rtol=1e-13)
    assert np.allclose(O._logpdf_halfcauchy(x, 0.1), st.halfcauchy(scale=0.1).logpdf(x), rtol=1e-13)
    assert np.allclose(O._logpdf_uniform(x, 0.01, 5), st.uniform(0.01, 4.99).logpdf(x), rtol=1e-13)
    assert np.allclose(O._logpdf_gamma(x, 2.0, 3.0), st.gamma(a=2.0, scale=1 / 3.0).logpdf(x), rtol=1e-13)
    ls = np.array([0.2, 1.5])
    want = (st.lognorm(s=1.0).logpdf(2.0) + st.halfcauchy(scale=0.1).logpdf(0.5)
            + np.sum(st.halfcauchy(scale=1.0).logpdf(1.0 / (0.5 * ls ** 2))))
    assert O.saas_prior_logprob(ls, 2.0, 0.5) == pytest.approx(want, rel=1e-13)
    X, y = ref_test_data(20, 2)
    gp = O.OracleGP(X, y, noise=1e-6, lengthscale_prior="DSLP")
    want = np.sum(st.lognorm(s=math.sqrt(3), scale=math.exp(math.sqrt(2) + 0.5 * math.log(2))).logpdf(ls)) \
        + st.uniform(1e-4, 1e8 - 1e-4).logpdf(2.0)
    assert gp.prior_logprob(ls, 2.0, 1.0) == pytest.approx(want, rel=1e-13)


@pytest.mark.parametrize("prior", [None, "DSLP", "SAAS"])
def test_neg_mll_value_and_grad_with_priors_fd(prior):
    X, y = ref_test_data(25, 2)
    gp = O.OracleGP(X, y, noise=1e-6, lengthscale_prior=prior)
    th = np.log(gp.get_hyperparams()) + 0.1
    f, g = gp.neg_mll_value_and_grad(th)
    assert f == pytest.approx(gp.neg_mll(th), rel=1e-12)
    for i in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[i] += 1e-5
        tm[i] -= 1e-5
        assert g[i] == pytest.approx((gp.neg_mll(tp) - gp.neg_mll(tm)) / 2e-5, rel=1e-4, abs=1e-6)


@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_fantasy_literal_equals_rank1(kernel):
    rng = np.random.default_rng(5)
    n, d, M, C = 60, 3, 17, 23
    X = rng.uniform(size=(n, d))
    y = np.cos(4 * X[:, 0]) + X[:, 1]
    gp = O.OracleGP(X, y, noise=1e-6, kernel=kernel, lengthscales=[0.3, 0.5, 0.4], kernel_variance=1.2)
    Z, cand = rng.uniform(size=(M, d)), rng.uniform(size=(C, d))
    cand[0] = X[3]                                 # candidate on top of a training point -> s_c ~ noise
    wv, ws = O.wip_sweep_literal(gp, cand, Z)
    r = O.wip_sweep(gp, cand, Z)
    assert np.max(np.abs(r["wipv"] - wv)) <= 1e-9 * gp.y_std ** 2
    assert np.max(np.abs(r["wipstd"] - ws)) <= 1e-8 * gp.y_std
    assert r["argmin_v"] == int(np.argmin(wv)) and r["argmin_s"] == int(np.argmin(ws))
    m, v = gp.predict_batched(cand)
    assert np.allclose(r["mean"], m, rtol=0, atol=1e-12) and np.allclose(r["var"], v, rtol=0, atol=1e-12)


def test_reference_gp_invariants():
    # reference tests/test_gp.py:129-139 — variance at a training point < 1e-3 (noise 1e-6, ls=1, kvar=1, N=25, d=2)
    X, y = ref_test_data(25, 2)
    gp = O.OracleGP(X, y, noise=1e-6)
    assert gp.predict_var_single(X[0]) < 1e-3
    assert gp.predict_var_single(np.array([0.5, 0.5])) > 0
    # GP interpolation mu(X) ~ y
    assert np.allclose(gp.predict_mean_batched(X), y.ravel(), atol=5e-3)
    # update adds two points and rejects a duplicate (tests/test_gp.py:165-169)
    n0 = gp.npoints
    gp.update(np.array([[0.11, 0.93], [0.87, 0.07]]), np.array([[-0.3], [-0.2]]))
    assert gp.npoints == n0 + 2
    gp.update(X[:1], y[:1])
    assert gp.npoints == n0 + 2
    # RBF vs Matern differ by > 1 % at (0.5, 0.5) (tests/test_gp.py:295)
    g1 = O.OracleGP(X, y, noise=1e-6, kernel="rbf", lengthscales=[0.2, 0.2])
    g2 = O.OracleGP(X, y, noise=1e-6, kernel="matern", lengthscales=[0.2, 0.2])
    m1, m2 = g1.predict_mean_single([0.5, 0.5]), g2.predict_mean_single([0.5, 0.5])
    assert m1 != m2


def test_fit_improves_mll_and_matches_restart_recipe():
    X, y = ref_test_data(40, 2)
    gp = O.OracleGP(X, y, noise=1e-6)
    rng = np.random.default_rng(7)
    x0 = O.restart_points(np.log(gp.get_hyperparams()), gp.hyperparam_bounds, 3, rng)
    assert x0.shape == (3, 3) and np.allclose(x0[0], 0.0)
    assert np.all(x0[1:] >= gp.hyperparam_bounds[0]) and np.all(x0[1:] <= gp.hyperparam_bounds[1])
    before = -gp.neg_mll(x0[0])
    res = gp.fit(x0=x0, maxiter=100)
    assert res["mll"] >= before
    gp.update_hyperparams(res["params"])
    assert np.all(np.isfinite(gp.cholesky))


def test_ei_logei():
    u = np.array([-50.0, -5.0, -1.0, -0.5, 0.0, 0.7, 3.0])
    direct = st.norm.pdf(u) + u * st.norm.cdf(u)
    assert np.allclose(O.ei_helper(u), direct, rtol=1e-12, atol=0)
    le = O.log_ei_helper(u)
    good = u > -30
    assert np.allclose(le[good], np.log(direct[good]), rtol=1e-9, atol=1e-9)
    # deep tail: asymptotic log EI ~ log phi(u) - 2 log|u|
    assert le[0] == pytest.approx(st.norm.logpdf(-50.0) - 2 * math.log(50.0), rel=1e-3)
    assert np.all(np.isfinite(O.log_ei_helper(np.array([-1e7, -1e6, -999999.0]))))
    assert np.all(O.ei_score(np.array([0.1, -2.0]), np.array([0.5, 1e-30]), 0.3) >= 0)     # tests/test_acquisition.py:92-95
    x = np.array([1e-3, 0.5, 0.7, 5.0])
    assert np.allclose(O.log1mexp(x), np.log(1 - np.exp(-x)), rtol=1e-12)


# ---- the plain-C second restatement against the NumPy one (two independent CPU implementations) ----
@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_c_restatement_agrees_with_numpy_restatement(kernel):
    from oracle import c_binding as OC
    kid = 0 if kernel == "rbf" else 1
    rng = np.random.default_rng(12)
    n, d = 45, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2]
    ls, kv, noise = np.array([0.35, 0.6, 0.8]), 1.7, 1e-6
    og = O.OracleGP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kv)
    ys = og.train_y.ravel()
    # kernel matrix, factor, alpha, MLL value
    kfun = O.get_kernel(kernel)
    assert np.allclose(OC.kernel(kid, X, X[:7], ls, kv, noise, False), kfun(X, X[:7], ls, kv, noise, include_noise=False),
                       rtol=1e-14, atol=0)
    info, mll, grad, L, alpha = OC.mll(kid, X, ys, ls, kv, noise)
    assert info == 0
    assert np.allclose(L, og.cholesky, atol=1e-11) and np.allclose(alpha, og.alphas.ravel(), rtol=1e-6)
    K = kfun(X, X, ls, kv, noise, include_noise=True)
    assert mll == pytest.approx(O.gp_mll(K, ys, n), rel=1e-12)
    # analytic gradient (C: explicit inverse, scalar loops) vs the NumPy analytic gradient
    _, g_np = O.mll_value_and_grad(kernel, X, ys, ls, kv, noise)
    assert np.allclose(grad, g_np, rtol=1e-8, atol=1e-10)
    # posterior and the literal (N+1)-factor fantasy variance
    Xq, Z = rng.uniform(size=(9, d)), rng.uniform(size=(11, d))
    m_c, v_c = OC.predict(kid, X, L, alpha, ls, kv, noise, Xq)
    m_np, v_np = og.predict_batched(Xq)
    assert np.allclose(m_c, m_np, atol=1e-10) and np.allclose(v_c, v_np, rtol=1e-7, atol=1e-12)
    for x in (Xq[0], X[3]):                       # a generic point and one on top of a training point
        f_c = OC.fantasy_var(kid, X, L, ls, kv, noise, x, Z, og.y_std)
        f_np = og.fantasy_var(x, Z, og._k12(Z))
        assert np.allclose(f_c, f_np, rtol=1e-6, atol=1e-12 * og.y_std ** 2)


def test_c_restatement_under_address_sanitizer():
    """the host-side C of the checker under -fsanitize=address,undefined (`make -C oracle asan`).
    The sanitizer runtime has to be the first library of the process, so the C oracle's tests run in a child
    interpreter with libasan preloaded; any report aborts the child."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    odir = os.path.join(os.path.dirname(here), "oracle")
    subprocess.run(["make", "-C", odir, "asan"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=asan, BOBE_ORACLE_C_LIB=os.path.join(odir, "libbobe_oracle_c_asan.so"),
               BOBE_ORACLE_XP_LIB=os.path.join(odir, "libbobe_oracle_xp_asan.so"), BOBE_XP_SKIP_QUAD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k",
                        "c_restatement_agrees or c_restatement_not_positive or extended_precision_truth"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
    assert "passed" in p.stdout


def test_c_restatement_not_positive_definite_is_nan():
    from oracle import c_binding as OC
    X = np.array([[0.1, 0.2], [0.1, 0.2], [0.7, 0.3]])
    info, mll, grad, L, alpha = OC.mll(0, X, np.array([1.0, -1.0, 0.0]), np.array([0.5, 0.5]), 1.0, 0.0)
    assert info > 0 and np.isnan(mll) and np.all(np.isnan(grad)) and np.all(np.isnan(L)) and np.all(np.isnan(alpha))


# ---- host logic: the first-order optimisers restated from optax (bobe_amd/optim.py), against torch.optim -------
@pytest.mark.parametrize("name,kw", [("adam", {}), ("sgd", {}), ("sgd", {"momentum": 0.9})])
def test_first_order_optimisers_follow_torch(name, kw):
    import torch
    from bobe_amd.optim import optimize_optax, optimize_optax_vmap

    A = np.array([[3.0, 0.5, 0.0], [0.5, 2.0, 0.3], [0.0, 0.3, 1.0]])
    c = np.array([0.7, 0.2, 0.4])

    def vg(x):
        r = np.asarray(x) - c
        return float(0.5 * r @ A @ r + 0.1 * np.sum(np.sin(3 * x))), A @ r + 0.3 * np.cos(3 * np.asarray(x))

    x0 = np.array([[0.1, 0.9, 0.5], [0.8, 0.1, 0.2]])
    lr, steps = 0.05, 40
    # independent trajectory with torch (unit bounds [0, 1]: the scaling is the identity, clipping after each step)
    finals, bests = [], []
    for row in x0:
        p = torch.tensor(row, dtype=torch.float64, requires_grad=True)
        opt = torch.optim.Adam([p], lr=lr, eps=1e-8) if name == "adam" else torch.optim.SGD([p], lr=lr, **kw)
        best = np.inf
        for _ in range(steps):
            f, g = vg(p.detach().numpy())
            best = min(best, f)
            p.grad = torch.tensor(g)
            opt.step()
            with torch.no_grad():
                p.clamp_(0.0, 1.0)
        finals.append(p.detach().numpy().copy())
        bests.append(best)
    i = int(np.argmin(bests))
    options = {"name": name, "lr": lr, "early_stop_patience": 1000, **kw}
    x, f = optimize_optax(vg, num_params=3, bounds=[0.0, 1.0], x0=x0, optimizer_options=dict(options), maxiter=steps,
                          n_restarts=2)
    assert f == pytest.approx(bests[i], rel=1e-12) and np.allclose(x, finals[i], atol=1e-12)
    xv, fv = optimize_optax_vmap(vg, num_params=3, bounds=[0.0, 1.0], x0=x0, optimizer_options=dict(options),
                                 maxiter=steps, n_restarts=2, batch_value_and_grad=lambda xs: [vg(q) for q in xs])
    assert fv == pytest.approx(bests[i], rel=1e-12)
    xw, fw = optimize_optax_vmap(vg, (), {}, 3, [0.0, 1.0], x0, dict(options), steps, 2)     # the reference's positional order
    assert fw == fv and np.array_equal(xw, xv)
    with pytest.raises(ValueError):
        optimize_optax(vg, (), {}, 3, [0.0, 1.0], x0, {"name": "lbfgs"}, 5, 2)


def test_svm_decision_restatement_against_sklearn():
    """oracle.bobe_oracle_loop.svm_predict is clf.py:188-213 line for line (direct differences); scikit-learn's own
    decision_function (libsvm, the expanded form) must agree to a few ulp of the terms' magnitude, and the 0/1
    probabilities wherever the decision is not inside that band."""
    from sklearn.svm import SVC
    from oracle import bobe_oracle_loop as OL
    for d, n, seed in ((2, 120, 0), (6, 400, 1), (10, 600, 2)):
        rng = np.random.default_rng(seed)
        X = rng.uniform(size=(n, d))
        y = -2000.0 * np.sum((X - 0.5) ** 2, axis=1)
        labels = np.where(y < y.max() - 75.0 * d, 0, 1)
        clf = SVC(kernel="rbf", gamma="scale", C=1e7).fit(X, labels)
        sv, dual, b, gam = clf.support_vectors_, clf.dual_coef_[0], float(clf.intercept_[0]), float(clf._gamma)
        q = np.vstack([rng.uniform(size=(1500, d)), X])
        ref = clf.decision_function(q)
        mine = OL.svm_predict(q, sv, dual, b, gam)
        scale = OL.svm_decision_scale(q, sv, dual, gam)
        tol = 1e-9 * np.abs(ref) + 1e-13 * scale
        assert np.all(np.abs(mine - ref) <= tol)
        clear = np.abs(ref) > tol
        assert np.array_equal(OL.svm_predict_proba(q, sv, dual, b, gam)[clear], (ref >= 0).astype(float)[clear])
        assert np.array_equal(clf.predict(q)[clear], (ref >= 0).astype(int)[clear])


@pytest.mark.parametrize("case", ["rbf_n50_d2", "matern_n130_d3", "rbf_n257_d5_saas"])
def test_oracle_against_sklearn_gpr(case):
    """One more independent implementation of the published algorithm (Rasmussen & Williams, alg. 2.1):
    scikit-learn's GaussianProcessRegressor with ``ConstantKernel x RBF(length_scale=ls)`` / ``x Matern(nu=2.5)``,
    ``alpha = noise``, no optimiser - on the inputs of the committed golden cases (tests/golden/make_golden.py).
    Compared: the log marginal likelihood and its gradient in log-space theta (what ``neg_mll`` differentiates,
    gp.py:385-398 / optim.py:306-309), the posterior mean and variance (gp.py:450-474).  This covers Matern-5/2 and
    mean / variance, which the notebook pin (tests/golden/reference_held.json) does not; it is a third-party
    implementation, not the reference: the oracle stays "parity unpinned" beyond that pin.
    scikit-learn's predictive variance leaves the noise out (its ``alpha`` sits on the training diagonal only); the
    reference's includes it (kernel_diag(..., include_noise=True), gp.py:463) - added here before comparing."""
    import os
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", case + ".npz"), allow_pickle=True)
    X, y, ls = g["X"], g["y"], g["lengthscales"]
    kvar, noise, kernel = float(g["kernel_variance"]), float(g["noise"]), str(g["kernel"])
    og = O.OracleGP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    ys = np.asarray(og.train_y).reshape(-1)                     # standardised targets (gp.py:283-307)
    base = RBF(length_scale=ls) if kernel == "rbf" else Matern(length_scale=ls, nu=2.5)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(kvar) * base, alpha=noise, optimizer=None, normalize_y=False)
    gpr.fit(X, ys)
    # --- LML and gradient at the fitted theta and at a perturbed one (theta = log [kvar, ls...] in scikit-learn's order)
    for shift in (0.0, 0.11):
        ls_t = ls * np.exp(shift * np.cos(np.arange(len(ls))))
        kv_t = kvar * math.exp(-shift)
        lml, grad = gpr.log_marginal_likelihood(np.log(np.append(kv_t, ls_t)), eval_gradient=True)
        mll, g_or = O.mll_value_and_grad(kernel, X, ys, ls_t, kv_t, noise)      # d/dlog ls_j ..., d/dlog kvar last
        assert mll == pytest.approx(lml, rel=1e-10)
        assert np.allclose(np.append(g_or[-1], g_or[:-1]), grad, rtol=1e-7, atol=1e-8 * np.max(np.abs(grad)))
        assert O.gp_mll(O.get_kernel(kernel)(X, X, ls_t, kv_t, noise, include_noise=True), ys.reshape(-1, 1), len(ys)) \
            == pytest.approx(lml, rel=1e-10)
    # --- posterior mean / variance at the golden candidates (one of them a training point)
    cand = g["cand"]
    mu, sd = gpr.predict(cand, return_std=True)
    assert np.allclose(og.predict_mean_batched(cand), mu * og.y_std + og.y_mean, rtol=0, atol=1e-8 * np.max(np.abs(y)))
    var_sk = (sd ** 2 + noise) * og.y_std ** 2
    var_or = np.asarray(og.predict_var_batched(cand)).reshape(-1)
    # (scikit-learn clips negative variances to 0 before the root; the reference floors at 1e-12 in standardised units)
    assert np.allclose(var_or, np.maximum(var_sk, 1e-12 * og.y_std ** 2), rtol=1e-6, atol=2e-7 * (kvar + noise) * og.y_std ** 2)
    # the frozen fixture says the same
    assert np.allclose(g["pred_mean"], mu * og.y_std + og.y_mean, rtol=0, atol=1e-8 * np.max(np.abs(y)))


@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_extended_precision_truth_agrees_with_the_oracle(kernel):
    """oracle/bobe_oracle_xp.c (the same quantities in x87 long double and in __float128: the reference point of
    tests/test_gpu_conditioning.py) on a WELL-conditioned case, where fp64 is accurate too: LML, gradient, posterior mean and
    variance, the variance at the integration points, the fantasy variance and its cross term must be the NumPy oracle's to
    fp64 rounding, and the two extended types must agree with each other far below that; a matrix that is not positive
    definite is reported, not factorised."""
    import os
    from oracle import c_binding as OC
    rng = np.random.default_rng(5)
    n, d = 90, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - X[:, 2]
    ls, kvar, noise = np.array([0.4, 0.6, 0.5]), 1.3, 1e-6
    og = O.OracleGP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    ys = np.asarray(og.train_y).reshape(-1)
    cand, Z = rng.uniform(size=(7, d)), rng.uniform(size=(5, d))
    kid = 0 if kernel == "rbf" else 1
    tr = OC.gp_truth(kid, X, ys, ls, kvar, noise, cand, Z)
    assert tr["info"] == 0 and tr["digits"] >= 64
    mll, g = O.mll_value_and_grad(kernel, X, ys, ls, kvar, noise)
    assert tr["mll"] == pytest.approx(mll, rel=1e-11) and np.allclose(tr["grad"], g, rtol=1e-8, atol=1e-9 * np.max(np.abs(g)))
    assert np.allclose(tr["mean"] * og.y_std + og.y_mean, og.predict_mean_batched(cand), rtol=0, atol=1e-10)
    assert np.allclose(tr["var"] * og.y_std ** 2, np.asarray(og.predict_var_batched(cand)).ravel(), rtol=1e-7, atol=1e-12)
    f = np.array([og.fantasy_var(c, Z, og._k12(Z)) for c in cand])
    assert np.allclose(tr["fantasy"] * og.y_std ** 2, f, rtol=1e-6, atol=1e-12)
    assert np.allclose(tr["var_z"] * og.y_std ** 2, np.asarray(og.predict_var_batched(Z)).ravel(), rtol=1e-7, atol=1e-12)
    assert np.allclose(tr["var_z"][None, :] - tr["cross"] ** 2 / tr["var"][:, None], tr["fantasy"], rtol=1e-9, atol=1e-13)
    assert 0 < tr["min_pivot"] <= kvar + noise
    if not os.environ.get("BOBE_XP_SKIP_QUAD"):           # (the sanitizer run has the long-double build only)
        tq = OC.gp_truth(kid, X, ys, ls, kvar, noise, cand, Z, kind="xq")
        assert tq["digits"] == 113 and abs(tq["mll"] - tr["mll"]) <= 1e-14 * abs(tq["mll"])
        assert np.allclose(tq["grad"], tr["grad"], rtol=1e-12, atol=0) and np.allclose(tq["fantasy"], tr["fantasy"], rtol=1e-11)
    Xd = X.copy()
    Xd[1] = Xd[0]                                           # a duplicated point without noise: singular
    bad = OC.gp_truth(kid, Xd, ys, ls, kvar, 0.0, want_grad=False)
    assert bad["info"] > 0
