"""The module-level functions of the reference's gp.py on the GPU (rows a1, a5, a15 of SURVEY.md 8a as FREE functions:
``dist_sq`` gp.py:80-96, ``gp_mll`` gp.py:170-178, ``fast_update_cholesky`` gp.py:181-197), the rank test's switch
(``bobe_gp_set_pivot_floor_ulp``, deviation (vii) of DESIGN.md 8) and the unmodified ``BOBE(...).run()`` call."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n1,n2,d", [(1, 1, 1), (7, 300, 3), (129, 64, 8), (260, 257, 20)])
def test_dist_sq_against_the_oracle(n1, n2, d):
    from bobe_amd.gp import dist_sq
    from oracle import bobe_oracle as O
    rng = np.random.default_rng(n1 + n2)
    x, y = rng.normal(size=(n1, d)), rng.normal(size=(n2, d))
    got = dist_sq(x, y)
    ref = O.dist_sq(x, y)
    assert got.shape == (n1, n2)
    assert np.allclose(got, ref, rtol=1e-14, atol=1e-14 * d)
    assert np.all(dist_sq(x, x).diagonal() == 0.0)


@pytest.mark.parametrize("n", [1, 2, 50, 129, 700])
def test_gp_mll_of_a_caller_supplied_matrix(n):
    """gp_mll(k, train_y, num_points) == the oracle's (LAPACK dpotrf / cho_solve) on the same matrix, 1e-10 relative."""
    from bobe_amd.gp import gp_mll, rbf_kernel
    from oracle import bobe_oracle as O
    rng = np.random.default_rng(n)
    X = rng.uniform(size=(n, 3))
    y = (np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - X[:, 2]).reshape(-1, 1)        # (smooth: y^T K^-1 y is not ~ 1 / noise)
    K = O.rbf_kernel(X, X, np.array([0.3, 0.5, 0.4]), 1.7, 1e-6, include_noise=True)
    ref = O.gp_mll(K, y, n)
    got = gp_mll(K, y, n)
    assert abs(got - ref) <= 1e-10 * abs(ref)
    # the same through the GPU-assembled kernel matrix (the call chain of GP.neg_mll, gp.py:385-398)
    got2 = gp_mll(rbf_kernel(X, X, np.array([0.3, 0.5, 0.4]), 1.7, 1e-6, include_noise=True), y, n)
    assert abs(got2 - ref) <= 1e-10 * abs(ref)


def test_gp_mll_not_positive_definite_is_nan():
    from bobe_amd.gp import gp_mll
    K = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert np.isnan(gp_mll(K, np.ones((2, 1)), 2))
    with pytest.raises(ValueError):
        gp_mll(np.eye(3), np.ones(2), 3)


@pytest.mark.parametrize("n", [1, 5, 128, 300])
def test_fast_update_cholesky_rank_one_append(n):
    """fast_update_cholesky(L, k, k_self) == the Cholesky factor of the bordered matrix (gp.py:181-197)."""
    from bobe_amd.gp import fast_update_cholesky
    from oracle import bobe_oracle as O
    rng = np.random.default_rng(10 + n)
    X = rng.uniform(size=(n + 1, 2))
    K = O.rbf_kernel(X, X, np.array([0.4, 0.6]), 1.3, 1e-6, include_noise=True)
    L = sla.cholesky(K[:n, :n], lower=True)
    got = fast_update_cholesky(L, K[:n, n], K[n, n])
    ref_o = O.fast_update_cholesky(L, K[:n, n], K[n, n])
    ref = sla.cholesky(K, lower=True)
    assert got.shape == (n + 1, n + 1) and np.array_equal(got[:n, :n], L) and np.all(np.triu(got, 1) == 0.0)
    scale = 1e-9 * np.max(np.abs(ref))
    assert np.allclose(got, ref_o, rtol=0, atol=scale) and np.allclose(got, ref, rtol=0, atol=scale)
    # k_self too small: a negative radicand gives NaN, as jnp.sqrt does
    assert np.isnan(fast_update_cholesky(L, K[:n, n], -1.0)[n, n])


def test_prior_helpers_of_the_reference_module():
    from bobe_amd.gp import DummyDistribution, make_distribution, saas_prior_logprob
    from oracle import bobe_oracle as O
    assert float(np.sum(DummyDistribution().log_prob(3.0))) == 0.0
    ln = make_distribution({"name": "LogNormal", "loc": 0.3, "scale": 1.2})
    assert float(ln.log_prob(0.7)) == pytest.approx(float(O.make_distribution({"name": "LogNormal", "loc": 0.3, "scale": 1.2})
                                                         (0.7)), rel=1e-13)
    ls = np.array([0.2, 0.9, 1.4])
    assert saas_prior_logprob(ls, 1.3, 0.4) == pytest.approx(float(O.saas_prior_logprob(ls, 1.3, 0.4)), rel=1e-12)


def test_default_is_the_reference_rule_and_the_rank_test_is_opt_in(caplog):
    """The drop-in class keeps the reference's positive-definiteness rule (gp.py:175, 549: ``jnp.linalg.cholesky`` fails on a
    pivot <= 0 only, like LAPACK's dpotrf): ``GP(...).pivot_floor_ulp == 0`` with no environment variable.  Deviation (vii),
    the rank test - a positive pivot below 64 ulp of k(x,x) + noise is NOT_PD - is what ``BOBE(...)`` opts into
    (``pivot_floor_ulp=64``).  Band between the two rules: both outcomes are exercised on one matrix, and the default's
    result agrees with LAPACK's dpotrf on the same K."""
    import logging
    from bobe_amd import GP
    from oracle import bobe_oracle as O
    assert "BOBE_PIVOT_FLOOR_ULP" not in os.environ
    rng = np.random.default_rng(3)
    n, d = 100, 2
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1]
    ls = np.array([1.5, 1.5])
    # walk the kernel variance up until the rank test refuses while LAPACK still factors
    found = None
    for kvar in 2.0 ** np.arange(10, 32):
        K = O.rbf_kernel(X, X, ls, kvar, 1e-8, include_noise=True)
        try:
            Lref = sla.cholesky(K, lower=True)
        except sla.LinAlgError:
            break
        gp = GP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=float(kvar), pivot_floor_ulp=64.0)
        if gp.not_pd:
            found = (float(kvar), Lref)
            break
    assert found is not None, "no kernel variance in the band between the rank test and LAPACK's sign test"
    kvar, Lref = found
    th = np.log(np.append(ls, kvar))
    # ---- the default: the reference's rule
    plain = GP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar)
    assert plain.pivot_floor_ulp == 0.0 and not plain.not_pd and np.all(np.isfinite(plain.cholesky))
    # what the sign-only rule lets through is as good as LAPACK's factor of the same matrix: compare the reconstructions
    K = O.rbf_kernel(X, X, ls, kvar, 1e-8, include_noise=True)
    Lg = plain.cholesky
    err_gpu = np.max(np.abs(Lg @ Lg.T - K)) / kvar
    err_lapack = np.max(np.abs(Lref @ Lref.T - K)) / kvar
    assert err_gpu <= max(8 * err_lapack, n * np.finfo(float).eps)          # (backward-stable: ||L L^T - K|| <= c n eps ||K||)
    f = plain.neg_mll(th)
    og = O.OracleGP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar)
    # in this band the log-determinant is built on pivots of rounding noise: the two fp64 factorisations differ in the
    # fourth digit, so each is measured against the extended-precision value (oracle/bobe_oracle_xp.c)
    from oracle import c_binding as CB
    tr = CB.gp_truth(0, X, np.asarray(og.train_y).reshape(-1), ls, kvar, 1e-8, want_grad=False)
    lp = float(plain.prior_func(ls, kvar))
    err_hip, err_lap = abs(-f - lp - tr["mll"]), abs(-og.neg_mll(th) - lp - tr["mll"])
    assert tr["info"] == 0 and np.isfinite(f) and err_hip <= 4 * err_lap + 1e-10 * abs(tr["mll"]), (err_hip, err_lap)
    # the batch / slot paths take the handle's setting too
    fb = plain.neg_mll_value_and_grad_batch([th, th], want_grad=False)
    assert all(np.isfinite(v[0]) for v in fb)
    # ---- opted in: refused, logged once per GP, NaN everywhere; switchable on a live GP; travels with copy()
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="bobe_amd"):
        strict = GP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar, pivot_floor_ulp=64.0)
        strict.recompute_cholesky()
    assert strict.pivot_floor_ulp == 64.0 and strict.not_pd and np.all(np.isnan(strict.cholesky))
    assert sum("rank test" in r.getMessage() for r in caplog.records) == 1
    assert np.isnan(strict.neg_mll(th))
    strict.pivot_floor_ulp = 0.0
    strict.recompute_cholesky()
    assert not strict.not_pd and np.array_equal(strict.cholesky, plain.cholesky)
    strict.pivot_floor_ulp = 64.0
    assert np.isnan(strict.neg_mll(th))
    with pytest.raises(Exception):
        strict.pivot_floor_ulp = -1.0
    assert strict.copy().pivot_floor_ulp == 64.0 and plain.copy().pivot_floor_ulp == 0.0


def test_large_kernel_variance_at_the_default_noise_gives_lapacks_mll():
    """``GP`` at kernel variance 3.5e5 and the reference's default noise of 1e-8 on a clustered 10-D design (the regime of the
    config-5 fits; with the 64-ulp rank test on, the top decades of ``kernel_variance_bounds`` = [1e-4, 1e8], gp.py:202, were
    NaN): a finite log marginal likelihood wherever LAPACK's dpotrf factors the matrix, as close to the extended-precision
    value as LAPACK's own."""
    from bobe_amd import GP
    from oracle import bobe_oracle as O
    from oracle import c_binding as CB
    assert "BOBE_PIVOT_FLOOR_ULP" not in os.environ
    rng = np.random.default_rng(11)
    n, d = 700, 10
    centre = np.full(d, 0.6)
    X = np.clip(np.vstack([rng.uniform(size=(200, d)), centre + 0.05 * rng.standard_normal((n - 200, d))]), 0.0, 1.0)
    y = -np.sum((X - centre) ** 2, axis=1) * 40.0
    ls = np.full(d, 3.0)
    for kvar in (3.5e5, 3.0e7):
        K = O.rbf_kernel(X, X, ls, kvar, 1e-8, include_noise=True)
        try:
            sla.cholesky(K, lower=True)
            lapack_ok = True
        except sla.LinAlgError:
            lapack_ok = False
        gp = GP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar)
        refused = GP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar, pivot_floor_ulp=64.0).not_pd
        if not lapack_ok:
            continue                                               # (beyond 1 / eps either factorisation may fail: no claim)
        assert not gp.not_pd, kvar
        th = np.log(np.append(ls, kvar))
        f = gp.neg_mll(th)
        og = O.OracleGP(X, y, noise=1e-8, lengthscales=ls, kernel_variance=kvar)
        tr = CB.gp_truth(0, X, np.asarray(og.train_y).reshape(-1), ls, kvar, 1e-8, want_grad=False)
        lp = float(gp.prior_func(ls, kvar))
        err_hip, err_lap = abs(-f - lp - tr["mll"]), abs(-og.neg_mll(th) - lp - tr["mll"])
        # (cond K is beyond 1 / eps here: either fp64 factorisation is whole units off the extended-precision value - what is
        # asserted is that the GPU's finite answer is no further from it than LAPACK's)
        assert np.isfinite(f) and err_hip <= 4 * err_lap + 1e-10 * abs(tr["mll"]), (kvar, refused, err_hip, err_lap)


def test_sampler_entry_points_refuse_device_pointers():
    """bobe_gp_rwalk / bobe_gp_hmc_run document host pointers; a device pointer comes back as BOBE_ERR_ARG, not as a
    hipMemcpy of the wrong kind."""
    import torch
    from bobe_amd import GP, _lib
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(30, 2))
    gp = GP(X, np.sum(X, axis=1), noise=1e-6)
    lib = _lib.load()
    P = 4
    xs = np.ascontiguousarray(rng.uniform(size=(P, 2)))
    logl = np.zeros(P)
    step = np.ascontiguousarray(0.1 * np.eye(2))
    nacc, nin = np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32)
    dev = torch.tensor(xs, device="cuda")
    st = lib.bobe_gp_rwalk(gp._h, P, C.c_void_p(dev.data_ptr()), _lib.ptr(logl), _lib.ptr(step), -1e30, 3, 1, 1.0, 0.0,
                           C.c_void_p(nacc.ctypes.data), C.c_void_p(nin.ctypes.data), None)
    assert st == -1 and "host pointers" in _lib.last_error()
    st = lib.bobe_gp_rwalk(gp._h, P, _lib.ptr(xs), _lib.ptr(logl), _lib.ptr(step), -1e30, 3, 1, 1.0, 0.0,
                           C.c_void_p(nacc.ctypes.data), C.c_void_p(nin.ctypes.data), None)
    assert st == 0


def test_unmodified_run_call_is_wipstd_in_batches_of_four(tmp_path, monkeypatch):
    """``BOBE(f, names, bounds).run()`` with no arguments: the reference's defaults (bo.py:967-984) - WIPStd, batches of 4,
    NUTS integration points, files written beside the run (save=True, save_dir='.')."""
    from bobe_amd.bo import BOBE
    monkeypatch.chdir(tmp_path)
    sig = 0.12

    def gauss(x):
        return -0.5 * float(np.sum(((np.asarray(x) - 0.5) / sig) ** 2))
    b = BOBE(gauss, ["a", "b"], np.array([[0.0, 1.0], [0.0, 1.0]]).T, seed=11, verbosity="WARNING")
    assert b.gp.npoints == 16 and (tmp_path / "loglikelihood_gp.npz").exists()      # (the reference's default name)
    res = b.run()
    assert b.acquisition.name.lower() == "wipstd" and b.batch_size == 4 and b.mc_points_method == "NUTS"
    assert (res["gp"].npoints - 16) % 4 <= 3 and len(res["acq_history"]) >= (200 - 16) // 4      # min_evals = 200 first
    assert res["termination_reason"] in ("LogZ converged", "Maximum evaluations reached", "Maximum GP size reached")
    assert res["termination_reason"] == "LogZ converged"
    truth = 2 * 0.5 * np.log(2 * np.pi * sig ** 2)
    assert abs(res["logz"]["mean"] - truth) < 0.1
    assert (tmp_path / "loglikelihood_run.json").exists()
    like = res["likelihood"]                                                        # a Likelihood, as in the reference
    assert like.name == "loglikelihood" and like.param_list == ["a", "b"] and like.param_bounds.shape == (2, 2)
    assert like([0.5, 0.5]) == 0.0 and like(np.array([[0.5, 0.5]])) == 0.0
    # the reference's eight result keys (bo.py:827-836) come first
    assert list(res)[:8] == ["gp", "likelihood", "results_manager", "best_val", "best_pt", "logz", "termination_reason",
                             "samples"]
    # the helper methods a script can call (bo.py:621, 681, 707, 758)
    n0 = b.gp.npoints
    pts, vals = b.get_next_batch({"mc_samples": b.mc_samples, "mc_points_size": 32}, n_batch=2, n_restarts=1, maxiter=20,
                                 early_stop_patience=5, step=999, verbose=False)
    new = b.evaluate_likelihood(pts, 999, verbose=False)
    assert pts.shape == (2, 2) and new.shape == (2, 1)
    b.update_gp(pts, new, step=999, verbose=False)
    assert n0 <= b.gp.npoints <= n0 + 2
    assert b.check_max_evals_and_gpsize(10 ** 9) and b.termination_reason == "Maximum evaluations reached"
