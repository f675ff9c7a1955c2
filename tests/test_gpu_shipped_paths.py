"""Parity tests for code paths that ship but that the smaller parity cases never reach (round-2 verdict, weak #2):

* the 128-tile variants of the inverse / K^-1-gradient kernels, selected from N = 6272 up (gp_factor.hip, ``lauum`` /
  ``trtri``): value, gradient and entries of K^-1 against the oracle's LAPACK evaluation (dpotrf / dpotri), both kernels;
* Matern through the lock-step batch (N >= 1024: per-slot hyper-parameter arrays in the assembly and gradient kernels):
  bitwise the single evaluation, and against the oracle;
* Matern at BASELINE config 2's size through the sweep;
* BASELINE config 3 (headline): 4096 candidates and the sweep's own argmin candidate against ``oracle.wip_sweep``.

Reference lines: gp.py:124-178 (kernels, gp_mll), optim.py:306-309 (value_and_grad), gp.py:552-576 and
acquisition.py:385-398, 438-465 (fantasy variance, WIPV / WIPStd, argmin).  Tolerances: SURVEY.md 8d.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bobe_oracle as O  # noqa: E402


def GP(*a, **k):
    from bobe_amd import GP as _GP
    return _GP(*a, **k)


def noisy_data(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X[:, 0]) + np.cos(2.0 * X[:, -1]) * X[:, d // 2] + 0.05 * rng.normal(size=n)
    return X, y


@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_value_gradient_and_kinv_through_the_128_tile_kernels(kernel):
    """N = 6400 = 50 blocks: 50*51/2 = 1275 >= 1200 selects k_lauum_grad<.,.,128> and the top inverse levels run
    k_trtri_T/R<128> (gp_factor.hip: lauum(), trtri()).  The largest gradient check before this one was N = 4096."""
    from bobe_amd import _lib
    from scipy.linalg import lapack
    n, d = 6400, 8
    X, y = noisy_data(n, d, seed=61)
    ls = 0.5 * (1.0 + 0.03 * np.arange(d))
    kv, noise = 1.2, 1e-5
    gp = GP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kv)
    assert not gp.not_pd and gp._lib.bobe_gp_npoints(gp._h) == n
    ys = gp.train_y.ravel()
    mll_cpu, g_cpu = O.cycle_value_and_grad(X, ys, ls, kv, noise, kernel=kernel)
    mll_gpu, g_gpu = gp.mll_data(ls, kv)
    assert abs(mll_gpu - mll_cpu) <= 1e-10 * abs(mll_cpu), (mll_gpu, mll_cpu)
    assert np.max(np.abs(g_gpu - g_cpu)) <= 1e-8 * np.max(np.abs(g_cpu)), (g_gpu, g_cpu)
    # the same evaluation on a slot and in a lock-step batch: identical bits (the tile size depends on N only)
    m_s, g_s = gp.mll_data(ls, kv, slot=1)
    assert m_s == mll_gpu and np.array_equal(g_s, g_gpu)
    mb, gb = gp.mll_data_batch(np.vstack([ls, 1.1 * ls]), np.array([kv, 0.9 * kv]))
    assert mb[0] == mll_gpu and np.array_equal(gb[0], g_gpu)
    m1, g1 = gp.mll_data(1.1 * ls, 0.9 * kv)
    assert mb[1] == m1 and np.array_equal(gb[1], g1)
    # K^-1 itself: 256 random entries (and the diagonal) against dpotri.  Forward error of an inverse is bounded by
    # cond(K) eps |K^-1|, cond(K) <= (N kvar + noise) / noise = 7.7e8 here -> 1e-6 max|K^-1| leaves a factor ten
    K = O.get_kernel(kernel)(X, X, ls, kv, noise, include_noise=True)
    Lo, info = lapack.dpotrf(K, lower=1, clean=1, overwrite_a=0)
    assert info == 0
    Ki_o, info = lapack.dpotri(Lo, lower=1, overwrite_c=1)
    assert info == 0
    Ki = np.empty((n, n))
    assert gp._lib.bobe_debug_kinv(gp._h, _lib.ptr(Ki)) == 0
    rng = np.random.default_rng(5)
    ii, jj = rng.integers(0, n, 256), rng.integers(0, n, 256)
    lo_i, lo_j = np.maximum(ii, jj), np.minimum(ii, jj)                  # dpotri filled the lower triangle
    scale = np.max(np.abs(np.diag(Ki_o)))
    assert np.max(np.abs(Ki[ii, jj] - Ki_o[lo_i, lo_j])) <= 1e-6 * scale
    assert np.max(np.abs(np.diag(Ki) - np.diag(Ki_o))) <= 1e-6 * scale
    assert np.array_equal(Ki[ii, jj], Ki[jj, ii])
    # the inverse factor behind it (128-tile recursive levels): Linv K Linv^T = I on a random row set
    Li = np.empty((n, n))
    assert gp._lib.bobe_debug_linv(gp._h, _lib.ptr(Li)) == 0
    rows = np.sort(rng.choice(n, 48, replace=False))
    R = Li[rows] @ K @ Li[rows].T
    assert np.max(np.abs(R - np.eye(48))) <= 1e-7


@pytest.mark.parametrize("n", [1024, 2048])
def test_matern_in_the_lockstep_batch(n):
    """bobe_gp_mll_batch from 1024 points up = ONE launch sequence for the B evaluations, every kernel reading its
    slot's hyper-parameters from a device array (k_kernel_matrix<1,...>, k_lauum_grad<1,...> with hp != nullptr)."""
    d, B = 6, 4
    X, y = noisy_data(n, d, seed=n)
    gp = GP(X, y, noise=1e-6, kernel="matern", lengthscales=np.full(d, 0.7), kernel_variance=1.1)
    rng = np.random.default_rng(n + 1)
    ls = np.exp(rng.uniform(np.log(0.3), np.log(1.5), size=(B, d)))
    kv = np.exp(rng.uniform(-0.4, 0.4, size=B))
    chol0 = gp.cholesky.copy()
    mb, gb = gp.mll_data_batch(ls, kv)
    ys = gp.train_y.ravel()
    for b in range(B):
        m1, g1 = gp.mll_data(ls[b], kv[b])
        assert mb[b] == m1 and np.array_equal(gb[b], g1), b                       # same bits as the lone evaluation
        mo, go = O.cycle_value_and_grad(X, ys, ls[b], kv[b], 1e-6, kernel="matern")
        assert abs(mb[b] - mo) <= 1e-10 * abs(mo)
        assert np.max(np.abs(gb[b] - go)) <= 1e-8 * np.max(np.abs(go))
    mv, none = gp.mll_data_batch(ls, kv, want_grad=False)
    assert none is None and np.array_equal(mv, mb)
    assert np.array_equal(gp.cholesky, chol0)                                     # the factored state is left alone
    # RBF and Matern handles do not share anything: the same batch on an RBF handle differs
    gr = GP(X, y, noise=1e-6, kernel="rbf", lengthscales=np.full(d, 0.7), kernel_variance=1.1)
    assert not np.allclose(gr.mll_data_batch(ls, kv)[0], mb)


def test_matern_sweep_at_config2_size():
    """BASELINE config 2's shape (N=1024, d=6, C=8192, M=512) with the Matern-5/2 kernel against oracle.wip_sweep."""
    from bobe_amd.synthetic import synthetic_problem
    N, d, Cn, M = 1024, 6, 8192, 512
    X, y, cand, Z = synthetic_problem(N, d, Cn, M)
    ls, kv, noise = np.full(d, 0.8) * (1 + 0.05 * np.arange(d)), 1.3, 1e-6
    gp = GP(X, y, noise=noise, kernel="matern", lengthscales=ls, kernel_variance=kv)
    og = O.OracleGP(X, y, noise=noise, kernel="matern", lengthscales=ls, kernel_variance=kv)
    r = gp.wip_sweep(cand, Z, want_mean_var=True)
    ro = O.wip_sweep(og, cand, Z)
    kself = kv + noise
    assert np.max(np.abs(r["mean"] - ro["mean"])) <= 1e-8 * np.max(np.abs(og.train_y))
    assert np.all(np.abs(r["var"] - ro["var"]) <= 1e-9 * kself + 1e-7 * np.abs(ro["var"]))
    s2 = og.y_std ** 2
    assert np.all(np.abs(r["wipv"] - ro["wipv"]) <= s2 * 1e-9 * kself + 1e-7 * ro["wipv"])
    assert np.all(np.abs(r["wipstd"] - ro["wipstd"]) <= og.y_std * 1e-9 + 1e-7 * ro["wipstd"])
    for key, sc in (("argmin_v", ro["wipv"]), ("argmin_s", ro["wipstd"])):
        best = int(np.argmin(sc))
        assert r[key] == best or abs(sc[r[key]] - sc[best]) <= 1e-9 * abs(sc[best])


def test_headline_sample_against_the_oracle_sweep():
    """BASELINE config 3 (N=4096, d=8, C=65 536, M=512): the full GPU sweep, then 4096 of its candidates plus the
    candidate it picked against oracle.wip_sweep (acquisition.py:385-398 / gp.py:552-576 restated) — what bench.py's
    cpu_baseline leg used to be the only place to check."""
    from bobe_amd.synthetic import synthetic_problem, theta_schedule
    N, d, Cn, M = 4096, 8, 65536, 512
    X, y, cand, Z = synthetic_problem(N, d, Cn, M)
    th = theta_schedule(d)
    ls, kv, noise = np.exp(th[-1, :d]), float(np.exp(th[-1, d])), 1e-6
    gp = GP(X, y, noise=noise, lengthscales=ls, kernel_variance=kv)
    r = gp.wip_sweep(cand, Z, want_mean_var=True)
    og = O.OracleGP(X, y, noise=noise, lengthscales=ls, kernel_variance=kv)
    rng = np.random.default_rng(11)
    idx = np.unique(np.concatenate([rng.choice(Cn, 4096, replace=False), [r["argmin_v"], r["argmin_s"]]]))
    ro = O.wip_sweep(og, cand[idx], Z)
    kself = kv + noise
    s2 = og.y_std ** 2
    assert np.max(np.abs(r["mean"][idx] - ro["mean"])) <= 1e-8 * np.max(np.abs(og.train_y))
    assert np.all(np.abs(r["var"][idx] - ro["var"]) <= 1e-9 * kself + 1e-7 * np.abs(ro["var"]))
    assert np.all(np.abs(r["wipv"][idx] - ro["wipv"]) <= s2 * 1e-9 * kself + 1e-7 * ro["wipv"])
    assert np.all(np.abs(r["wipstd"][idx] - ro["wipstd"]) <= og.y_std * 1e-9 + 1e-7 * ro["wipstd"])
    # argmin: exact on the sample (both scores), and the sweep's pick over ALL candidates is the sample's best too
    assert int(np.argmin(r["wipv"][idx])) == int(np.argmin(ro["wipv"]))
    assert int(np.argmin(r["wipstd"][idx])) == int(np.argmin(ro["wipstd"]))
    assert idx[int(np.argmin(ro["wipv"]))] == r["argmin_v"]
    assert idx[int(np.argmin(ro["wipstd"]))] == r["argmin_s"]
    assert r["argmin_v"] == int(np.argmin(r["wipv"])) and r["argmin_s"] == int(np.argmin(r["wipstd"]))


def test_few_candidate_gradient_path_equals_the_batched_one_at_large_n():
    """bobe_gp_wip_grad takes matrix-vector stages spread over the chip for up to 16 candidates (the acquisition's L-BFGS
    refinement sends one) and 128-column tile passes above: the same scores and gradients, at sizes where the first has
    hundreds of partial sums per candidate (k_wg_rows: one training row per wave)."""
    from bobe_amd import GP
    for n, d, m in ((3000, 7, 300), (130, 3, 64)):
        rng = np.random.default_rng(1)
        X = rng.uniform(size=(n, d))
        gp = GP(X, -10 * np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.7), kernel_variance=2.0)
        Z, c = rng.uniform(size=(m, d)), rng.uniform(size=(20, d))
        big = gp.wip_grad(c, Z)
        one = [gp.wip_grad(c[i:i + 1], Z) for i in range(20)]
        five = gp.wip_grad(c[:5], Z)
        for k in range(4):
            scale = np.max(np.abs(big[k]))
            assert np.max(np.abs(np.concatenate([f[k] for f in one]) - big[k])) < 1e-8 * scale
            assert np.max(np.abs(five[k] - big[k][:5])) < 1e-8 * scale


@pytest.mark.parametrize("n,d,m,kernel,kappa", [(300, 3, 96, "rbf", -1.0), (300, 3, 96, "matern", 0.0), (700, 10, 256, "rbf", 0.0),
                                                (1000, 6, 130, "rbf", -1.0), (4096, 4, 2048, "rbf", -1.0), (4096, 4, 2048, "rbf", 0.0)])
def test_sweeping_the_integration_points_themselves_reuses_their_solve_bit_for_bit(n, d, m, kernel, kappa):
    """acquisition.py:394 sweeps the integration points themselves (candidates = mc_points).  The library then reuses V_Z = L^-1
    K(X, Z) as the candidates' V (no second assembly and solve) and takes the column sums from it in the association the
    candidates' solve would have used (k_colsq_tile_parts): the scores must be the BITS of the long way - which the same call
    takes when the posterior mean / variance are asked for as well - for both forms of the solve (plain product / blocked
    substitution), both tilings of the Z side (64 x 64 tiles for few integration points, 128 x 128 above) and ragged sizes."""
    from bobe_amd import GP
    rng = np.random.default_rng(n + m)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) - np.sum((X - 0.4) ** 2, axis=1)
    gp = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=np.full(d, 0.5), kernel_variance=1.3)
    gp.refine_kappa = kappa
    gp.recompute_cholesky()
    assert gp.refining == (kappa == 0.0)
    Z = rng.uniform(size=(m, d))
    short = gp.wip_sweep(Z, Z)                                   # the shortcut
    long_ = gp.wip_sweep(Z, Z, want_mean_var=True)               # mean / variance asked for: assembly + solve of the candidates
    other = gp.wip_sweep(Z.copy() + 0.0, np.array(Z))           # equal content in other buffers: still the shortcut
    for k in ("wipv", "wipstd"):
        assert np.array_equal(short[k], long_[k]) and np.array_equal(short[k], other[k]), k
    assert short["argmin_v"] == long_["argmin_v"] and short["argmin_s"] == long_["argmin_s"]
    fa = gp.fantasy_var(Z[:40], Z[:40])                          # (candidates == integration points here as well)
    fb = gp.fantasy_var(Z[:40], np.vstack([Z[:40], Z[:1]]))[:, :40]
    assert np.allclose(fa, fb, rtol=1e-9, atol=1e-13)
