"""The blocked forward substitution that replaces the products with the inverse factor where the factor is ill conditioned
(include/bobe_gp.h: bobe_gp_set_refine_kappa, bobe_gp_set_solve_block), entry point by entry point.

Where it matters - the reference's default noise of 1e-8 with large kernel variances - is tests/test_gpu_conditioning.py.
Here it is FORCED ON (kappa = 0) for a well-conditioned GP, where the plain product is already accurate: every entry
point that forms v = L^-1 k (predict, sweep, fantasy_var, wip_grad on both of its paths, predict_grad, the rank-b append)
must return what it returns with the plain product, up to rounding, and the oracle's values.  That exercises the launch
sequence of bobe_gp::solve_v (k_blk_step: long panel updates, diagonal solves, short updates inside a panel) at several
block / panel heights, the vector form of the few-candidate path (one refinement step) and the standalone cross launch on
ragged sizes, both kernels and more than one candidate chunk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, -1]
    return rng, X, y


@pytest.mark.parametrize("block,panel", [(128, 512), (128, 128), (128, 256), (256, 256), (256, 1024), (384, 512)])
@pytest.mark.parametrize("n,d,kernel", [(130, 2, "rbf"), (641, 5, "matern"), (1000, 3, "rbf")])
def test_forced_substitution_changes_nothing_beyond_rounding(n, d, kernel, block, panel):
    """diagonal blocks of `block` rows, `panel` rows per long update launch (bobe_gp_set_solve_block, bobe_debug_solve_opts)"""
    from bobe_amd import GP
    from oracle import bobe_oracle as O
    rng, X, y = _problem(n, d, n)
    ls = np.full(d, 0.45)
    plain = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.4)
    assert plain.refine_kappa == 1e6 and plain.solve_block == 128       # the defaults
    plain.refine_kappa = -1.0                                           # never: the plain product with the inverse factor
    plain.recompute_cholesky()
    assert not plain.refining
    forced = GP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.4)
    forced.refine_kappa = 0.0
    forced.solve_block = block
    assert forced.solve_block == block
    assert forced._lib.bobe_debug_solve_opts(forced._h, panel, 0) == 0
    forced.recompute_cholesky()
    assert forced.refining and np.array_equal(forced.cholesky, plain.cholesky)
    og = O.OracleGP(X, y, noise=1e-6, kernel=kernel, lengthscales=ls, kernel_variance=1.4)
    cand, Z = rng.uniform(size=(700, d)), rng.uniform(size=(96, d))
    cand[3] = X[5]                                                          # a training point among the candidates
    for gp in (plain, forced):
        gp._lib.bobe_gp_set_chunk(gp._h, 256)                               # three chunks: fused and standalone cross launches
    a, b = plain.wip_sweep(cand, Z, want_mean_var=True), forced.wip_sweep(cand, Z, want_mean_var=True)
    ro = O.wip_sweep(og, cand, Z)
    scale_v = og.y_std ** 2 * (1.4 + 1e-6)
    for k, tol in (("mean", 1e-11), ("var", 1e-11 * 1.4), ("wipv", 1e-10 * scale_v), ("wipstd", 1e-10 * np.sqrt(scale_v))):
        assert np.allclose(a[k], b[k], rtol=1e-9, atol=tol), k
        assert np.allclose(b[k], ro[k], rtol=1e-6, atol=1e3 * tol), k
    assert a["argmin_s"] == b["argmin_s"] == ro["argmin_s"] and a["argmin_v"] == b["argmin_v"] == ro["argmin_v"]
    fa, fb = plain.fantasy_var(cand[:9], Z), forced.fantasy_var(cand[:9], Z)
    assert np.allclose(fa, fb, rtol=1e-8, atol=1e-10 * scale_v)
    m1, v1 = plain.predict_batched(cand[:300])
    m2, v2 = forced.predict_batched(cand[:300])
    assert np.allclose(m1, m2, rtol=0, atol=1e-11) and np.allclose(v1, v2, rtol=1e-9, atol=1e-11 * 1.4)
    for idx in (np.arange(2), np.arange(20)):                               # few-candidate path / batched path
        ga, gb = plain.wip_grad(cand[idx], Z), forced.wip_grad(cand[idx], Z)
        for qa, qb in zip(ga, gb):
            assert np.allclose(qa, qb, rtol=1e-7, atol=1e-9 * scale_v)
    pa, pb = plain.predict_grad(cand[:40]), forced.predict_grad(cand[:40])
    for qa, qb in zip(pa, pb):
        assert np.allclose(qa, qb, rtol=1e-7, atol=1e-9)


def test_refinement_through_update_append_and_copy():
    """GP.update at unchanged hyper-parameters is a rank-b append (bobe_gp_append) whose new rows are L^-1 K(X_old, X_new):
    with the substitution forced on they must still be the rows a fresh factorisation gives, the setting must survive the append
    and travel with copy()."""
    from bobe_amd import GP
    rng, X, y = _problem(300, 3, 9)
    ls = np.full(3, 0.5)
    gp = GP(X[:290], y[:290], noise=1e-6, lengthscales=ls, kernel_variance=1.2)
    gp.refine_kappa = 0.0
    gp.recompute_cholesky()
    assert gp.refining
    gp.update(X[290:], y[290:].reshape(-1, 1))                              # ten rows appended
    assert gp.npoints == 300 and gp.refining
    fresh = GP(X, y, noise=1e-6, lengthscales=ls, kernel_variance=1.2)
    assert np.allclose(gp.cholesky, fresh.cholesky, rtol=0, atol=1e-10)
    q = rng.uniform(size=(50, 3))
    (m1, v1), (m2, v2) = gp.predict_batched(q), fresh.predict_batched(q)
    assert np.allclose(m1, m2, rtol=0, atol=1e-9) and np.allclose(v1, v2, rtol=1e-7, atol=1e-11)
    c = gp.copy()
    assert c.refining and c.refine_kappa == 0.0
    c.refine_kappa = -1.0                                                   # never
    c.recompute_cholesky()
    assert not c.refining
    with pytest.raises(Exception):
        c.refine_kappa = float("nan")
    for bad in (0, 100, -128):
        with pytest.raises(Exception):
            c.solve_block = bad
    c.solve_block = 256
    assert c.copy().solve_block == 256


def test_default_threshold_switches_on_where_the_factor_is_ill_conditioned():
    """noise 1e-8 with a long length scale: (kvar + noise) / smallest pivot passes 1e6 and the substitution is on by itself; at
    noise 1e-6 with unit kernel variance (the headline configuration) it never is."""
    from bobe_amd import GP
    _, X, y = _problem(400, 2, 4)
    assert GP(X, y, noise=1e-8, lengthscales=np.full(2, 1.5), kernel_variance=50.0).refining
    assert not GP(X, y, noise=1e-6, lengthscales=np.full(2, 0.3), kernel_variance=1.0).refining
