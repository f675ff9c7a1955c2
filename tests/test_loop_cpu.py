"""Host logic either side of the hot path, on the CPU: the product's pure-Python pieces against the oracle's twins
(oracle/bobe_oracle_loop.py) — refit policy (bo.py:632-655), integration-point draw (acquisition.py:485-489), the
evidence integral (samplers.py:27-50) — and the frozen loop fixture against the oracle that made it."""
import os

import numpy as np
import pytest

from oracle import bobe_oracle as O
from oracle import bobe_oracle_loop as OL

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loop_himmelblau.npz")


def test_refit_policy_matches_oracle_twin_over_the_size_classes():
    from bobe_amd.bo import refit_policy
    for n in (2, 50, 199, 200, 201, 400, 749, 750, 751, 1200, 4096):
        for since in (0, 1, 3, 9, 39, 40):
            for n_new in (1, 2, 5):
                for fit_n_points in (1, 2, 4, 10, 50):
                    assert refit_policy(n, since, n_new, fit_n_points) == OL.refit_policy(n, since, n_new, fit_n_points)
    # the strict '<' of bo.py:639/644: N == 200 and N == 750 use the large-set branch (4 restarts, maxiter 200)
    assert OL.refit_policy(200, 0, 40, 10)[1:3] == (4, 200) and OL.refit_policy(750, 0, 40, 10)[1:3] == (4, 200)
    assert OL.refit_policy(199, 0, 2, 10)[:3] == (True, 8, 1000) and OL.refit_policy(201, 0, 9, 10)[0] is False


def test_get_mc_points_draws_the_oracles_indices():
    from bobe_amd.acquisition import get_mc_points
    x = np.random.default_rng(0).uniform(size=(300, 3))
    for seed, size in ((1, 64), (2, 300), (3, 17)):
        got = get_mc_points({"x": x}, mc_points_size=size, rng=np.random.default_rng(seed))
        want = O.get_mc_points(x, size, np.random.default_rng(seed))
        assert np.array_equal(got, want)
    with pytest.raises(IndexError):           # acquisition.py:486: n_samples < M indexes out of range, like the reference
        get_mc_points({"x": x[:10]}, mc_points_size=64, rng=np.random.default_rng(4))


def test_compute_integrals_matches_the_loop_twin_and_a_closed_form():
    from bobe_amd.samplers import compute_integrals
    rng = np.random.default_rng(5)
    n, nlive = 400, 50
    logl = np.sort(rng.normal(size=n) * 3.0)
    logvol = -np.arange(1, n + 1) / nlive
    for squared in (False, True):
        got = compute_integrals(logl=logl, logvol=logvol, squared=squared)
        want = OL.compute_integrals(logl, logvol, squared=squared)
        assert np.allclose(got, want, rtol=1e-13, atol=1e-13)
    rw = rng.normal(size=n)
    assert np.allclose(compute_integrals(logl, logvol, reweight=rw), OL.compute_integrals(logl, logvol, reweight=rw), rtol=1e-13)
    # constant likelihood L: the trapezoid sum telescopes to L (1 - X_n) apart from the first half-interval,
    # where the pad value exp(-1e300) = 0 stands in for L_0:  Z = L [ (1 - X_n) - (1 - X_1)/2 ]
    c = 1.7
    z = compute_integrals(np.full(n, c), logvol)[-1]
    x1, xn = np.exp(logvol[0]), np.exp(logvol[-1])
    assert z == pytest.approx(c + np.log((1 - xn) - 0.5 * (1 - x1)), abs=1e-12)


def test_logz_bounds_bracket_the_mean_and_shrink_with_the_variance():
    rng = np.random.default_rng(6)
    n, nlive = 300, 40
    logl = np.sort(rng.normal(size=n))
    logvol = -np.arange(1, n + 1) / nlive
    mean = OL.compute_integrals(logl, logvol)[-1]
    wide = OL.logz_bounds(logl, logvol, np.full(n, 0.25), mean)
    tight = OL.logz_bounds(logl, logvol, np.full(n, 1e-6), mean)
    assert wide["lower"] < mean < wide["upper"] and tight["lower"] < mean < tight["upper"]
    assert wide["upper"] - wide["lower"] == pytest.approx(1.0, abs=1e-9)          # logl +- 0.5 shifts logZ by +-0.5
    assert tight["upper"] - tight["lower"] < 1e-2 and tight["var"] < wide["var"]


def test_clf_gate_twin():
    m, v = np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2, 0.3])
    gm, gv = OL.clf_gate(m, v, np.array([0.9, 0.5, 0.49]), 0.5, -1e10)
    assert np.array_equal(gm, [1.0, 2.0, -1e10]) and np.array_equal(gv, [0.1, 0.2, 1e-12])
    assert OL.clf_gate(m, v, None, 0.5, -1e10)[0] is m or np.array_equal(OL.clf_gate(m, v, None, 0.5, -1e10)[0], m)
    assert np.array_equal(OL.clf_labels(np.array([0.0, -10.0, -300.0]), 250.0), [1, 1, 0])


def test_loop_fixture_is_what_the_oracle_produces():
    g = np.load(GOLD)
    it = 2
    gp = O.OracleGP(g[f"s{it}_train_x"], g[f"s{it}_train_y"], lengthscales=g[f"s{it}_lengthscales"],
                    kernel_variance=float(g[f"s{it}_kernel_variance"]))
    xs, acq, infos = OL.get_next_batch(gp, "wipstd", g[f"s{it}_mc_samples"], int(g["mc_points_size"]),
                                       int(g["n_batch"][it]), np.random.default_rng(1000 + it))
    assert np.allclose(xs, g[f"s{it}_batch_x"], atol=1e-9) and np.allclose(acq, g[f"s{it}_batch_val"], rtol=1e-9)
    assert [i["sweep_index"] for i in infos] == list(g[f"s{it}_sweep_index"])
    # believer members differ, and each refined point scores no worse than the sweep's pick (acquisition.py:403-412)
    assert len({tuple(np.round(x, 6)) for x in xs}) == len(xs)
    assert np.all(acq <= g[f"s{it}_sweep_value"] * (1 + 1e-12))
    # the refit schedule of the fixture: batch sizes 1,1,3,3,3 with threshold min(2, fit_n_points) = 2
    assert [bool(g[f"s{k}_refit"]) for k in range(int(g["n_iters"]))] == [False, True, True, True, True]


class _GaussSurface:
    """duck-typed surrogate with a known evidence: an isotropic Gaussian well inside the unit cube"""

    def __init__(self, d, s):
        self.ndim, self.s = d, s

    def predict_mean_batched(self, u):
        return -0.5 * np.sum(((np.atleast_2d(u) - 0.5) / self.s) ** 2, axis=1)

    def predict_var_batched(self, u):
        return np.full(np.atleast_2d(u).shape[0], 1e-12)


@pytest.mark.parametrize("d,s,method", [(2, 0.05, "ellipsoid"), (2, 0.05, "rwalk"), (6, 0.05, "rwalk"), (6, 0.05, "auto")])
def test_nested_sampler_recovers_a_known_evidence(d, s, method):
    """The host nested sampler (batched proposals, scored through predict_mean_batched) on an analytic surface:
    logZ = d log(s sqrt(2 pi)) for both proposal kinds, within three of its own error bars."""
    import math
    from bobe_amd import samplers
    truth = d * math.log(s * math.sqrt(2 * math.pi))
    smp, lz, ok = samplers.nested_sampling(_GaussSurface(d, s), ndim=d, mode="convergence", rng=np.random.default_rng(1),
                                           sample_method=method)
    assert ok and abs(lz["mean"] - truth) < 3 * lz["dlogz_sampler"] + 0.05
    assert lz["ncall"] < 2e6 and lz["lower"] <= lz["mean"] <= lz["upper"]
    w = smp["weights"] / smp["weights"].sum()
    assert np.allclose(np.sum(w[:, None] * smp["x"], axis=0), 0.5, atol=5 * s / math.sqrt(200))


class _CappedSurface(_GaussSurface):
    """a Gaussian well with its top cut off: once every live point sits on the plateau no replacement with L > L*
    exists, which is the sampler's give-up case"""

    def predict_mean_batched(self, u):
        return np.minimum(super().predict_mean_batched(u), -2.0)


@pytest.mark.parametrize("method", ["ellipsoid", "rwalk"])
def test_nested_sampler_terminations_count_every_point_once(method):
    """Every sample of a run is either a dead point or a final live point, never both (the worst point of the last
    iteration used to be retired AND kept); a run cut by maxcall says so; a run that could not find a replacement is
    not a successful one (bo.py must not call its logZ converged)."""
    from bobe_amd import samplers
    # (1) normal termination
    smp, lz, ok = samplers.nested_sampling(_GaussSurface(2, 0.05), ndim=2, rng=np.random.default_rng(3), sample_method=method)
    assert ok and not lz["truncated"]
    assert len(np.unique(smp["x"], axis=0)) == len(smp["x"]) == lz["niter"] + 500
    # (2) cut by maxcall: flagged, still every point once
    smp, lz, ok = samplers.nested_sampling(_GaussSurface(2, 0.05), ndim=2, rng=np.random.default_rng(3), sample_method=method,
                                           maxcall=20000)
    assert lz["truncated"] and len(np.unique(smp["x"], axis=0)) == len(smp["x"]) == lz["niter"] + 500
    # (3) plateau: nothing above L* exists any more -> gives up, unsuccessful, every point once
    smp, lz, ok = samplers.nested_sampling(_CappedSurface(2, 0.2), ndim=2, rng=np.random.default_rng(3), sample_method=method,
                                           nlive=100, batch=512)
    assert not ok and lz["truncated"]
    assert len(np.unique(smp["x"], axis=0)) == len(smp["x"]) == lz["niter"] + 100


class _BlackoutSurface(_GaussSurface):
    """the Gaussian well, except that 200 consecutive proposal batches score nothing above any threshold: the replacement
    search that meets them succeeds on its 201st refill"""

    def __init__(self, d, s, first, length=200):
        super().__init__(d, s)
        self.calls, self.first, self.length = 0, first, length

    def predict_mean_batched(self, u):
        self.calls += 1
        if self.first <= self.calls < self.first + self.length:
            return np.full(np.atleast_2d(u).shape[0], -1e300)
        return super().predict_mean_batched(u)


def test_replacement_found_on_the_last_refill_is_not_a_give_up():
    """A search that needs 201 refills and then finds its replacement has NOT given up: the run goes on, and when it later
    converges by dlogz it is a successful run (the give-up flag used to stay set for the rest of the run, so the next empty
    pool ended it as 'no acceptable replacement' and a converged logZ was thrown away)."""
    import math
    from bobe_amd import samplers
    surf = _BlackoutSurface(2, 0.05, first=4)
    smp, lz, ok = samplers.nested_sampling(surf, ndim=2, rng=np.random.default_rng(3), sample_method="ellipsoid", batch=256)
    assert surf.calls > 4 + 200                                    # the blackout was met and outlasted
    assert ok and not lz["truncated"]
    assert abs(lz["mean"] - 2 * math.log(0.05 * math.sqrt(2 * math.pi))) < 3 * lz["dlogz_sampler"] + 0.05
