"""Evidence consumer on the GPU GP: nested sampling of the surrogate + logZ bounds."""
import math

import numpy as np
import pytest
from scipy.stats import qmc

pytestmark = pytest.mark.gpu


def test_nested_sampling_gaussian_evidence_and_bounds():
    from bobe_amd import GP
    from bobe_amd.samplers import nested_sampling
    d, sig = 2, 0.15
    X = qmc.Sobol(d, scramble=True, seed=3).random(256)
    y = -0.5 * np.sum(((X - 0.5) / sig) ** 2, axis=1)
    gp = GP(X, y, noise=1e-8, lengthscales=[0.5, 0.5], kernel_variance=10.0)
    from bobe_amd.bo import gp_fit
    gp_fit(gp, maxiters=100, n_restarts=2, rng=np.random.default_rng(0))
    samples, logz, ok = nested_sampling(gp, mode="convergence", dlogz=0.01, rng=np.random.default_rng(1))
    analytic = d * 0.5 * math.log(2 * math.pi * sig ** 2)          # tails beyond 3.3 sigma are negligible
    assert ok and set(logz) >= {"mean", "dlogz_sampler", "upper", "lower", "var", "std"}
    assert logz["mean"] == pytest.approx(analytic, abs=0.2)
    assert logz["lower"] <= logz["mean"] <= logz["upper"]
    assert (logz["upper"] - logz["lower"]) / 2 < 0.5               # bo.py:886-891 convergence quantity is finite and small
    assert set(samples) == {"x", "weights", "logl", "best", "method"} and samples["method"] == "nested"
    assert np.all(np.abs(samples["best"] - 0.5) < 0.05)
    w = samples["weights"]
    post_mean = (w[:, None] * samples["x"]).sum(0) / w.sum()
    assert np.all(np.abs(post_mean - 0.5) < 0.03)
    # 'acq' mode: equal-weight samples usable as mc_samples for WIPV/WIPStd (acquisition.py:473-475)
    s2, _, _ = nested_sampling(gp, mode="acq", rng=np.random.default_rng(2))
    assert np.all(s2["weights"] == 1.0) and s2["x"].shape[1] == d
    assert np.all(np.abs(s2["x"].mean(0) - 0.5) < 0.05)


def test_bo_loop_with_logz_convergence():
    from bobe_amd.bo import BOBE
    sig = 0.2
    bounds = np.array([[0.0, 1.0], [0.0, 1.0]]).T
    bobe = BOBE(lambda x: -0.5 * float(np.sum(((x - 0.5) / sig) ** 2)), ["a", "b"], bounds, n_sobol_init=16, seed=5, save=False)
    res = bobe.run(acq="wipstd", max_evals=80, fit_n_points=4, batch_size=2, mc_points_size=64, mc_points_method="NS",
                   logz_threshold=0.05, min_evals=24, ns_n_points=8)
    assert "logz" in res and res["logz"]["mean"] == pytest.approx(2 * 0.5 * math.log(2 * math.pi * sig ** 2), abs=0.25)
    assert res["n_evals"] <= 80 and (res["converged"] or res["n_evals"] == 80)


def test_hmc_on_the_surrogate_recovers_a_gaussian_posterior():
    """sample_GP_NUTS (samplers.py:216-360): target, keywords and return dict of the reference; the sampler is a
    batched HMC (one bobe_gp_predict_grad call per leapfrog step for all chains)."""
    from bobe_amd import GP
    from bobe_amd.acquisition import get_mc_points, get_mc_samples
    from bobe_amd.bo import gp_fit
    from bobe_amd.samplers import get_hmc_settings, sample_GP_NUTS
    d = 3
    mu, sig = np.array([0.45, 0.55, 0.5]), np.array([0.08, 0.12, 0.1])
    X = qmc.Sobol(d, scramble=True, seed=5).random(512)
    y = -0.5 * np.sum(((X - mu) / sig) ** 2, axis=1)
    gp = GP(X, y, noise=1e-8, lengthscales=[0.5] * d, kernel_variance=10.0)
    gp_fit(gp, maxiters=100, n_restarts=2, rng=np.random.default_rng(0))
    assert get_hmc_settings(d) == (256, 1024, 4) and get_hmc_settings(12) == (512, 2048, 4)      # samplers.py:196-214
    s = sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), num_chains=4)
    assert set(s) == {"x", "logp", "best", "method"} and s["method"] == "MCMC"
    assert s["x"].shape == (4 * 1024 // 4, d) and s["logp"].shape == (1024,)
    assert np.all(s["x"] > 0) and np.all(s["x"] < 1)
    assert np.all(np.abs(s["x"].mean(0) - mu) < 0.02)
    assert np.all(np.abs(s["x"].std(0) / sig - 1.0) < 0.2)
    assert np.all(np.abs(s["best"] - mu) < 0.06)
    # logp is the (physical) GP mean at the samples
    assert np.allclose(s["logp"][:50], gp.predict_mean_batched(s["x"][:50]), atol=1e-6)
    # tempering widens the posterior: sigma scales like sqrt(temp)
    s4 = sample_GP_NUTS(gp, np_rng=np.random.default_rng(2), num_chains=2, temp=4.0, num_samples=512)
    assert s4["x"].shape[0] == 2 * 512 // 4
    assert np.all(s4["x"].std(0) > 1.5 * sig)
    # the reference's default mc_points_method (acquisition.py:466-471)
    mc = get_mc_samples(gp, warmup_steps=128, num_samples=256, thinning=4, method="NUTS", num_chains=4,
                        np_rng=np.random.default_rng(3))
    assert mc["x"].shape == (256, d)
    pts = get_mc_points(mc, mc_points_size=64, rng=np.random.default_rng(4))
    assert pts.shape == (64, d)


def test_fused_hmc_trajectory_equals_the_step_by_step_leapfrog():
    """bobe_gp_hmc_leapfrog (L leapfrog steps of every chain in one launch) against the same trajectory stepped on the
    host with one bobe_gp_predict_grad call per step, and against the oracle's posterior mean at the end point."""
    from scipy.special import expit
    from bobe_amd import GP
    from oracle import bobe_oracle as O
    for kernel, d in (("rbf", 3), ("matern", 5), ("rbf", 10)):
        rng = np.random.default_rng(d)
        X = rng.uniform(size=(150, d))
        y = -20.0 * np.sum((X - 0.45) ** 2, axis=1) + 3.0
        ls = np.linspace(0.3, 0.8, d)
        gp = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.7)
        og = O.OracleGP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.7)
        P, L, eps, temp = 7, 9, 0.07, 1.3
        U = rng.normal(size=(P, d))
        inv_mass = rng.uniform(0.5, 2.0, size=d)

        def lpg(Uq):
            Xq = np.clip(expit(Uq), 1e-12, 1 - 1e-12)
            m, _, dm, _ = gp.predict_grad(Xq, mean_only=True)
            mean = m * gp.y_std + gp.y_mean
            return (mean / temp + np.sum(np.log(Xq) + np.log1p(-Xq), axis=1),
                    dm * gp.y_std / temp * (Xq * (1 - Xq)) + (1 - 2 * Xq), mean, Xq)
        _, g0, _, _ = lpg(U)
        p0 = rng.normal(size=(P, d))
        Un, pn = U.copy(), p0 + 0.5 * eps * g0
        for s in range(L):
            Un = Un + eps * inv_mass * pn
            lpn, gn, meann, Xn = lpg(Un)
            pn = pn + (eps if s < L - 1 else 0.5 * eps) * gn
        Uf, pf, lpf, gf, meanf, Xf = gp.hmc_leapfrog(U, p0 + 0.5 * eps * g0, inv_mass, eps, L, temp)
        assert np.allclose(Uf, Un, rtol=1e-10, atol=1e-10) and np.allclose(pf, pn, rtol=1e-9, atol=1e-9)
        assert np.allclose(lpf, lpn, rtol=1e-10, atol=1e-9) and np.allclose(gf, gn, rtol=1e-8, atol=1e-8)
        assert np.allclose(Xf, Xn, rtol=1e-12) and np.allclose(meanf, meann, rtol=1e-10, atol=1e-9)
        assert np.allclose(meanf, og.predict_mean_batched(Xf), rtol=1e-8, atol=1e-7)


def test_hmc_sampler_same_statistics_fused_and_stepwise():
    from bobe_amd import GP
    from bobe_amd.samplers import sample_GP_NUTS
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(120, 2))
    y = -0.5 * np.sum(((X - np.array([0.4, 0.6])) / 0.12) ** 2, axis=1)
    gp = GP(X, y, noise=1e-6, lengthscales=[0.5, 0.5], kernel_variance=50.0)
    a = sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), warmup_steps=256, num_samples=1024, thinning=2)
    b = sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), warmup_steps=256, num_samples=1024, thinning=2,
                       fused_trajectories=False)
    c = sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), warmup_steps=256, num_samples=1024, thinning=2,
                       device_chains=False)
    for s in (a, b, c):                  # chains on the device / one call per leapfrog step / one call per trajectory
        assert np.allclose(s["x"].mean(0), [0.4, 0.6], atol=0.03) and np.allclose(s["x"].std(0), 0.12, atol=0.03)
        assert np.all(np.isfinite(s["logp"])) and s["best"].shape == (2,)
    assert a["x"].shape == b["x"].shape == c["x"].shape
    a2 = sample_GP_NUTS(gp, np_rng=np.random.default_rng(1), warmup_steps=256, num_samples=1024, thinning=2)
    assert np.array_equal(a["x"], a2["x"])                       # same seed, same chains


def test_warmup_step_size_gives_the_target_acceptance():
    """Dual averaging aims at 0.8 mean acceptance (NumPyro's default target).  Its averaged iterate is only meaningful
    if the averaging restarts with the metric at each mass-matrix window — counter included; with the counter left
    running the adopted step size came out as eps^0.73 (biased towards 1) and the sampling phase accepted far less."""
    from bobe_amd import GP
    from bobe_amd.samplers import sample_GP_NUTS
    d = 4
    mu, sig = np.full(d, 0.5), np.array([0.05, 0.08, 0.11, 0.07])
    X = qmc.Sobol(d, scramble=True, seed=9).random(768)
    y = -0.5 * np.sum(((X - mu) / sig) ** 2, axis=1)
    gp = GP(X, y, noise=1e-8, lengthscales=[0.6] * d, kernel_variance=50.0)
    diag = {}
    sample_GP_NUTS(gp, np_rng=np.random.default_rng(3), num_chains=4, warmup_steps=512, num_samples=64, diagnostics=diag)
    st, ad = diag["state"], diag["adapt"]
    acc = []
    for k in range(48):                                         # 48 more trajectories of the 64 chains, step size frozen
        _, _, dbg = gp.hmc_run(st, ad, diag["inv_mass"], seed=diag["seed"], it0=diag["it"] + k, niter=1, do_adapt=False,
                               debug=True)
        acc.append(dbg[:, d + 2])
    mean_acc = float(np.mean(acc))
    assert 0.68 <= mean_acc <= 0.92, mean_acc
    assert np.all(diag["eps"] > 1e-3) and np.all(diag["eps"] < 2.0)


def _hmc_state(gp, U, temp):
    from scipy.special import expit
    X = np.clip(expit(U), 1e-12, 1 - 1e-12)
    m, _, dm, _ = gp.predict_grad(X, mean_only=True)
    mean = m * gp.y_std + gp.y_mean
    lp = mean / temp + np.sum(np.log(X) + np.log1p(-X), axis=1)
    g = dm * gp.y_std / temp * (X * (1 - X)) + (1 - 2 * X)
    return np.ascontiguousarray(np.concatenate([U, g, X, lp[:, None], mean[:, None]], axis=1))


def test_device_chain_iteration_replayed_on_the_host():
    """One iteration of bobe_gp_hmc_run with its draws (momentum, L, uniform) read back: the same trajectory through
    bobe_gp_hmc_leapfrog and the Metropolis rule on the host must give the same acceptance probability and the same
    next state.  Then: launch boundaries and batch size do not change a chain (counter-based random numbers).  The sizes
    cover the three homes of a chain's training points in k_hmc_run (consumer_kernels.hpp, ChainRows): registers only,
    registers + LDS (N = 2500 at d = 12), and a streamed rest (N = 3000 at d = 12; N = 1500 at d = 20, the 32-wide variant)."""
    from bobe_amd import GP
    for kernel, d, n in (("rbf", 2, 200), ("matern", 6, 200), ("rbf", 12, 200), ("rbf", 6, 1500), ("matern", 12, 2500),
                         ("rbf", 12, 3000), ("matern", 20, 1500)):
        rng = np.random.default_rng(10 + d)
        X = rng.uniform(size=(n, d))
        y = -15.0 * np.sum((X - 0.5) ** 2, axis=1)
        gp = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=np.linspace(0.4, 0.9, d), kernel_variance=2.0)
        P, temp, eps = 24, 1.0, 0.05
        inv_mass = rng.uniform(0.5, 2.0, size=d)
        U0 = rng.normal(scale=0.5, size=(P, d))
        st = _hmc_state(gp, U0, temp)
        st0 = st.copy()
        adapt = np.tile(np.array([eps, 0.0, 0.0, 0.0, 0.0]), (P, 1))
        _, _, dbg = gp.hmc_run(st, adapt, inv_mass, seed=1234, it0=7, niter=1, do_adapt=False, temp=temp, debug=True)
        p0, L, r, ap = dbg[:, :d], dbg[:, d].astype(int), dbg[:, d + 1], dbg[:, d + 2]
        assert np.all((L >= 4) & (L <= 12)) and np.all((r > 0) & (r < 1))
        g0, lp0 = st0[:, d:2 * d], st0[:, 3 * d]
        for c in range(P):
            Un, pn, lpn, gn, meann, Xn = gp.hmc_leapfrog(U0[c:c + 1], p0[c:c + 1] + 0.5 * eps * g0[c:c + 1], inv_mass, eps,
                                                         int(L[c]), temp)
            h0 = lp0[c] - 0.5 * np.sum(p0[c] ** 2 * inv_mass)
            h1 = lpn[0] - 0.5 * np.sum(pn[0] ** 2 * inv_mass)
            ap_host = min(1.0, np.exp(h1 - h0)) if np.isfinite(h1) else 0.0
            assert ap[c] == pytest.approx(ap_host, rel=1e-9, abs=1e-12)
            want = np.concatenate([Un[0], gn[0], Xn[0], lpn, meann]) if r[c] < ap[c] else st0[c]
            assert np.allclose(st[c], want, rtol=1e-10, atol=1e-10)
        assert np.array_equal(adapt[:, 0], np.full(P, eps))                     # no adaptation asked for
        # 5 iterations in one launch == 3 + 2 in two launches == the first 8 chains alone
        a, b, c8 = st0.copy(), st0.copy(), st0[:8].copy()
        ad = lambda n: np.tile(np.array([eps, 0.0, 0.0, 0.0, 0.0]), (n, 1))
        aa, ab, ac = ad(P), ad(P), ad(8)
        gp.hmc_run(a, aa, inv_mass, 99, 0, 5, True, temp)
        gp.hmc_run(b, ab, inv_mass, 99, 0, 3, True, temp)
        gp.hmc_run(b, ab, inv_mass, 99, 3, 2, True, temp)
        gp.hmc_run(c8, ac, inv_mass, 99, 0, 5, True, temp)
        assert np.array_equal(a, b) and np.array_equal(aa, ab) and np.array_equal(a[:8], c8) and np.array_equal(aa[:8], ac)
        assert np.all(aa[:, 4] == 5) and not np.array_equal(aa[:, 0], np.full(P, eps))   # every chain adapted its step


def test_device_chain_random_numbers():
    """Momentum draws ~ N(0, 1 / inv_mass), trajectory lengths uniform on 4..12, Metropolis uniforms uniform."""
    from bobe_amd import GP
    rng = np.random.default_rng(3)
    d, P = 4, 4096
    X = rng.uniform(size=(50, d))
    gp = GP(X, -np.sum((X - 0.5) ** 2, axis=1), noise=1e-6, lengthscales=np.full(d, 0.7))
    inv_mass = np.array([0.5, 1.0, 2.0, 4.0])
    st = _hmc_state(gp, rng.normal(scale=0.3, size=(P, d)), 1.0)
    adapt = np.tile(np.array([0.05, 0.0, 0.0, 0.0, 0.0]), (P, 1))
    _, _, dbg = gp.hmc_run(st, adapt, inv_mass, seed=5, it0=0, niter=1, do_adapt=False, debug=True)
    p0, L, r = dbg[:, :d], dbg[:, d], dbg[:, d + 1]
    assert np.allclose(p0.mean(0), 0.0, atol=0.08) and np.allclose(p0.var(0) * inv_mass, 1.0, atol=0.08)
    assert abs(np.corrcoef(p0[:, 0], p0[:, 1])[0, 1]) < 0.05 and abs(np.corrcoef(p0[:-1, 0], p0[1:, 0])[0, 1]) < 0.05
    counts = np.bincount(L.astype(int), minlength=13)[4:]
    assert counts.sum() == P and np.all(np.abs(counts - P / 9) < 5 * np.sqrt(P / 9))
    assert abs(r.mean() - 0.5) < 0.02 and abs(r.var() - 1 / 12) < 0.01


def test_device_random_walks_of_nested_sampling():
    """bobe_gp_rwalk (all constrained Metropolis steps of all walkers in one launch) against the host rule, step by step:
    (i) one step with the proposals read back: accepted exactly where the proposal is inside the cube and the surrogate's
    mean there (bobe_gp_predict) exceeds L*; proposals are x + step z with z ~ N(0, I); (ii) many steps: every walker that
    moved ends inside the cube above L* with its own mean; a walker's path depends on (seed, walker) only; (iii) under a
    classifier gate no walker enters the infeasible region; (iv) nested sampling gets the same evidence with the walks on
    the device and stepped from the host."""
    from bobe_amd import GP, samplers
    from bobe_amd.clf_gp import GPwithClassifier
    for kernel, d, n in (("rbf", 3, 150), ("matern", 10, 1300), ("rbf", 20, 1600), ("matern", 12, 4000)):   # (registers;
        rng = np.random.default_rng(20 + d)                                   # registers + LDS; + a streamed rest)
        X = rng.uniform(size=(n, d))
        y = -40.0 * np.sum((X - 0.5) ** 2, axis=1)
        gp = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=np.full(d, 0.6), kernel_variance=2.0)
        P = 4096
        x0 = rng.uniform(0.05, 0.95, size=(P, d))
        l0 = gp.predict_mean_batched(x0)
        lstar = float(np.quantile(l0, 0.3))
        A = np.linalg.cholesky(np.cov(x0[:500], rowvar=False) + 1e-12 * np.eye(d)) * 0.35
        x1, l1, nacc, nin, prop = gp.rwalk(x0, l0, A, lstar, 1, seed=77, debug=True)
        inside = np.all((prop >= 0) & (prop <= 1), axis=1)
        assert np.array_equal(nin, inside.astype(np.int32)) and 0.3 < inside.mean() <= 1.0
        lp = np.full(P, -np.inf)
        lp[inside] = gp.predict_mean_batched(prop[inside])
        clear = np.abs(lp - lstar) > 1e-9 * max(1.0, abs(lstar))
        acc = inside & (lp > lstar)
        assert np.array_equal(nacc[clear] > 0, acc[clear]) and 0.1 < acc.mean() < 0.95
        assert np.array_equal(x1[acc & clear], prop[acc & clear]) and np.array_equal(x1[~acc & clear], x0[~acc & clear])
        assert np.allclose(l1[acc & clear], lp[acc & clear], rtol=1e-10, atol=1e-9) and np.array_equal(l1[~acc & clear], l0[~acc & clear])
        z = np.linalg.solve(A, (prop - x0).T).T                              # the draws behind the proposals
        assert np.allclose(z.mean(0), 0.0, atol=0.08) and np.allclose(np.cov(z, rowvar=False), np.eye(d), atol=0.1)
        # many steps
        x2, l2, na2, ni2 = gp.rwalk(x0, l0, A, lstar, 30, seed=5)
        moved = na2 > 0
        assert moved.mean() > 0.5 and np.all((x2 >= 0) & (x2 <= 1)) and np.all(l2[moved] > lstar)
        assert np.allclose(l2[moved], gp.predict_mean_batched(x2[moved]), rtol=1e-10, atol=1e-9)
        assert np.array_equal(x2[~moved], x0[~moved]) and np.all(na2 <= ni2) and np.all(ni2 <= 30)
        xa, la, _, _ = gp.rwalk(x0, l0, A, lstar, 30, seed=5)
        xb, lb, _, _ = gp.rwalk(x0[:100], l0[:100], A, lstar, 30, seed=5)
        assert np.array_equal(xa, x2) and np.array_equal(xb, x2[:100]) and np.array_equal(lb, l2[:100])
    # (iii) gated
    rng = np.random.default_rng(9)
    d = 2
    X = rng.uniform(size=(250, d))
    y = -800.0 * np.sum((X - np.array([0.45, 0.55])) ** 2, axis=1)
    g = GPwithClassifier(X, y, clf_threshold=40.0, gp_threshold=120.0, noise=1e-6, lengthscales=np.full(d, 0.3), minus_inf=-1e10)
    assert g.use_clf
    x0 = g.train_x_clf[g._clf_predict_func(g.train_x_clf) >= 0.5][:64]
    x0 = np.tile(x0, (8, 1))
    l0 = g.predict_mean_batched(x0)
    xg, lg, nag, _ = g.rwalk(x0, l0, 0.2 * np.eye(d), float(np.min(l0)) - 50.0, 40, seed=3)
    assert nag.sum() > 0 and np.all(g._clf_predict_func(xg) >= 0.5) and np.all(lg > g.minus_inf)
    # (iv) the sampler end to end, 6-D
    rng = np.random.default_rng(1)
    d = 6
    X = rng.uniform(size=(400, d))
    y = -30.0 * np.sum((X - 0.5) ** 2, axis=1)
    gp = GP(X, y, noise=1e-6, lengthscales=np.full(d, 0.8), kernel_variance=4.0)
    _, lz_dev, ok_dev = samplers.nested_sampling(gp, mode="convergence", dlogz=0.05, rng=np.random.default_rng(2), nlive=400)
    _, lz_host, ok_host = samplers.nested_sampling(gp, mode="convergence", dlogz=0.05, rng=np.random.default_rng(3), nlive=400,
                                                   device_walks=False)
    assert ok_dev and ok_host
    err = 4.0 * np.hypot(lz_dev["dlogz_sampler"], lz_host["dlogz_sampler"])
    assert abs(lz_dev["mean"] - lz_host["mean"]) < max(err, 0.3), (lz_dev, lz_host)
    exact = np.log((np.pi / 30.0) ** (d / 2))                                # integral of exp(-30 |x - 1/2|^2) over the cube
    assert abs(lz_dev["mean"] - exact) < 0.5 and abs(lz_host["mean"] - exact) < 0.5


def test_cross_lane_sums_of_the_sampler_kernels():
    """kernels_common.hpp: chain_wave_sum and wave_sum_components (v_permlane32/16_swap + DPP row_ror / half_mirror /
    quad_perm pairings instead of ds_bpermute).  Integer-valued inputs make every order of summation exact: the sums must
    equal NumPy's to the bit; with arbitrary doubles they agree to rounding; a single marked lane shows up in exactly its
    component (no lane is lost or counted twice)."""
    from bobe_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for D in (8, 16, 32):
        ints = np.ascontiguousarray(rng.integers(-1000, 1000, size=(64, D)).astype(np.float64))
        out = np.empty(2 * D)
        assert lib.bobe_debug_wave_sums(0, D, _lib.ptr(ints), _lib.ptr(out)) == 0, lib.bobe_last_error()
        assert np.array_equal(out[:D], ints.sum(0)) and np.array_equal(out[D:], ints.sum(0))
        vals = np.ascontiguousarray(rng.normal(size=(64, D)) * 10.0 ** rng.integers(-3, 4, size=(64, D)))
        assert lib.bobe_debug_wave_sums(0, D, _lib.ptr(vals), _lib.ptr(out)) == 0
        scale = np.abs(vals).sum(0)
        assert np.all(np.abs(out[:D] - vals.sum(0)) <= 1e-15 * scale) and np.all(np.abs(out[D:] - vals.sum(0)) <= 1e-15 * scale)
        for lane in (0, 1, 5, 17, 31, 32, 47, 63):
            one = np.zeros((64, D))
            one[lane] = np.arange(1, D + 1)
            assert lib.bobe_debug_wave_sums(0, D, _lib.ptr(np.ascontiguousarray(one)), _lib.ptr(out)) == 0
            assert np.array_equal(out[:D], np.arange(1, D + 1)) and np.array_equal(out[D:], np.arange(1, D + 1))
    assert lib.bobe_debug_wave_sums(0, 12, _lib.ptr(ints), _lib.ptr(out)) < 0            # only the three widths exist
