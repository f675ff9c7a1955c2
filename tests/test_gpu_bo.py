"""End-to-end BO loop on the GPU GP (reference tests/test_bo_2d.py:41-62, 112-136)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def himmelblau(x):   # reference tests/test_bo_2d.py: negative Himmelblau / 10-ish scale
    return -((x[0] ** 2 + x[1] - 11) ** 2 + (x[0] + x[1] ** 2 - 7) ** 2) / 10.0


def rosenbrock(x):
    return -((1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2) / 20.0


def test_wipstd_uniform_loop_himmelblau():
    from bobe_amd.bo import BOBE
    bounds = np.array([[-4.0, 4.0], [-4.0, 4.0]]).T
    bobe = BOBE(himmelblau, ["x", "y"], bounds, n_sobol_init=8, seed=1, save=False)
    res = bobe.run(acq="wipstd", max_evals=36, fit_n_points=2, batch_size=2, mc_points_size=64, num_mc_samples=256,
                   mc_points_method="uniform")
    assert set(res) >= {"gp", "best_val", "best_x", "n_evals", "acq_history", "timing"}
    assert res["best_val"] > -500                       # tests/test_bo_2d.py:143-170 style bounds
    assert 8 < res["gp"].npoints <= 36 and res["n_evals"] == res["gp"].npoints
    assert len(res["acq_history"]) >= 5 and all(np.isfinite(res["acq_history"]))
    assert res["timing"]["GP Training"] > 0 and res["timing"]["Acquisition Optimization"] > 0


def test_ei_loop_rosenbrock_improves():
    from bobe_amd.bo import BOBE
    bounds = np.array([[-1.0, 4.0], [-1.0, 7.0]]).T      # reference examples/Rosenbrock.py:25-28
    bobe = BOBE(rosenbrock, ["x", "y"], bounds, n_sobol_init=8, seed=3, save=False)
    start = float(np.max(bobe.gp.train_y * bobe.gp.y_std + bobe.gp.y_mean))
    res = bobe.run(acq="ei", max_evals=24)
    assert res["best_val"] >= start and res["best_val"] > -1000     # tests/test_bo_2d.py:69-97
    assert res["gp"].npoints <= 24


def test_gp_fit_restart_recipe_matches_oracle():
    from bobe_amd import GP
    from bobe_amd.bo import gp_fit
    from oracle import bobe_oracle as O
    rng = np.random.RandomState(42)
    X = rng.uniform(size=(40, 2))
    y = -np.sum((X - 0.5) ** 2, axis=1)
    gp = GP(X, y, noise=1e-6)
    og = O.OracleGP(X, y, noise=1e-6)
    r = gp_fit(gp, maxiters=50, n_restarts=3, rng=np.random.default_rng(5))
    x0 = O.restart_points(np.log(og.get_hyperparams()), og.hyperparam_bounds, 3, np.random.default_rng(5))
    ro = og.fit(x0=x0, maxiter=50)
    # same x0 recipe (pool.py:277-286) and the same optimiser; the optimum lies on a flat ridge in kvar,
    # so the two L-BFGS-B runs may stop a few 1e-7 apart in relative MLL
    assert r["mll"] == pytest.approx(ro["mll"], rel=1e-5)
    assert np.allclose(np.log(gp.get_hyperparams())[:2], ro["params"][:2], atol=5e-2)


def test_run_level_resume_continues_the_interrupted_run(tmp_path):
    """bo.py:327-381: ``BOBE(..., resume=True, resume_file=<save_dir>/<name>)`` rebuilds the run from ``<name>_gp.npz``
    (L and alpha restored without a factorisation) and the run state saved beside it.  The resumed run must CONTINUE
    the interrupted one: same kriging-believer picks as an uninterrupted run with the same seed (BASELINE config 1's
    likelihood and settings, shortened)."""
    import json
    import math
    from bobe_amd.bo import BOBE

    def banana(x):                                              # examples/Banana.py: curved 2-D likelihood
        return -0.25 * (5 * (0.2 - x[0])) ** 2 - (20 * (x[1] / 4 - x[0] ** 4)) ** 2

    bounds = np.array([[-1.0, 1.0], [-1.0, 2.0]]).T
    kw = dict(acq="wipstd", min_evals=100, fit_n_points=4, batch_size=2, mc_points_size=64, mc_points_method="uniform",
              num_mc_samples=512, logz_threshold=1e-6)

    def make(**extra):
        return BOBE(banana, ["x1", "x2"], bounds, n_sobol_init=8, seed=123, likelihood_name="banana", **extra)
    full = make(save=False).run(max_evals=20, **kw)                        # uninterrupted: 8 + 6 x 2 evaluations
    d1 = str(tmp_path / "run")
    first = make(save=True, save_dir=d1, save_step=1).run(max_evals=18, **kw)     # cut after 5 iterations
    assert first["n_evals"] == 18 and (tmp_path / "run" / "banana_gp.npz").exists()
    run_state = json.load(open(tmp_path / "run" / "banana_run.json"))
    # the run state names ONE complete generation of files; older generations are pruned once it is in place
    assert (tmp_path / "run" / run_state["gp_file"]).exists() and (tmp_path / "run" / run_state["mc_file"]).exists()
    assert sorted(f.name for f in (tmp_path / "run").iterdir()) == sorted(
        ["banana_gp.npz", "banana_run.json", run_state["gp_file"], run_state["mc_file"]])
    assert np.array_equal(first["gp"].train_x, full["gp"].train_x[:18])             # (saving does not disturb the run)
    # a kill between the files of the NEXT generation and its run state: a newer, unrelated <name>_gp.npz and a half-written
    # generation lie beside a run state that still names the old one - the resumed run must come up from the generation its
    # run state names, not from a mixture
    make(save=False).gp.save(str(tmp_path / "run" / "banana_gp"))                    # (an 8-point GP under the shared name)
    (tmp_path / "run" / "banana_gp.999.npz").write_bytes(b"half written")
    calls = []

    def counting(x):
        calls.append(np.array(x))
        return banana(x)
    again = BOBE(counting, ["x1", "x2"], bounds, n_sobol_init=8, seed=999, likelihood_name="banana", resume=True,
                 resume_file=str(tmp_path / "run" / "banana"), save=True, save_dir=d1, save_step=1)
    assert not again.fresh_start and again.gp.npoints == 18 and not calls          # no initial design is evaluated
    assert np.array_equal(again.gp.cholesky, first["gp"].cholesky)                  # the factor came from the file
    # (the file holds the targets in physical units, gp.py:597-634: the restored GP standardises them again - equal to rounding)
    assert np.allclose(again.gp.train_y, first["gp"].train_y, rtol=1e-13, atol=1e-15)
    assert math.isclose(again.gp.y_std, first["gp"].y_std, rel_tol=1e-14)
    res = again.run(max_evals=20, **kw)
    # iteration 6 of the resumed run = iteration 6 of the uninterrupted one: the same kriging-believer batch (same
    # integration samples, same generator state), the same refit.  (The restored L^-1 differs from the interrupted run's in
    # the last bits - it is rebuilt from L, there it came from appends - so "same" is to rounding, and this likelihood
    # with 20 points is ill-determined enough for later iterations to amplify that: only the first one is compared.)
    assert res["n_evals"] == 20 and len(calls) == 2
    assert np.allclose(res["gp"].train_x, full["gp"].train_x, atol=1e-7)
    assert len(res["acq_history"]) == len(full["acq_history"]) == 6
    assert np.allclose(res["acq_history"], full["acq_history"], rtol=1e-6, atol=1e-12)
    assert np.allclose(res["lengthscales"], full["lengthscales"], rtol=1e-5)
    assert math.isclose(res["kernel_variance"], full["kernel_variance"], rel_tol=1e-5)
    assert math.isclose(res["best_val"], full["best_val"], rel_tol=1e-9)
    more = BOBE(banana, ["x1", "x2"], bounds, n_sobol_init=8, seed=1, likelihood_name="banana", resume=True,
                resume_file=str(tmp_path / "run" / "banana"), save=False).run(max_evals=28, **kw)
    assert more["n_evals"] == 28 and len(more["acq_history"]) == 10                 # 6 iterations on file + 4 more
    # a GP file without run state resumes at iteration 0 with that training set; an unreadable one starts afresh
    (tmp_path / "run" / "banana_run.json").unlink()
    bare = BOBE(banana, ["x1", "x2"], bounds, n_sobol_init=8, seed=5, likelihood_name="banana", resume=True,
                resume_file=str(tmp_path / "run" / "banana"), save=False)
    assert not bare.fresh_start and bare.gp.npoints == 20 and bare._resume_state is None
    fresh = BOBE(banana, ["x1", "x2"], bounds, n_sobol_init=8, seed=5, likelihood_name="banana", resume=True,
                 resume_file=str(tmp_path / "nothing_here" / "banana"), save=False)
    assert fresh.fresh_start and fresh.gp.npoints == 8


def test_stages_of_a_tuple_acq_share_the_budgets_and_resume_in_place(tmp_path):
    """``run(acq=('ei', 'wipstd'))`` runs the acquisition functions as stages (bo.py:1149-1158).  A stage that would start past
    the budgets is not entered (round 5's driver evaluated one more batch there), a later stage starts with its own stopping
    flags, and the run state names the stage it was written in so that a resume continues THERE."""
    import json
    from bobe_amd.bo import BOBE
    bounds = np.array([[-4.0, 4.0], [-4.0, 4.0]]).T
    calls = []

    def counted(x):
        calls.append(1)
        return himmelblau(x)
    kw = dict(fit_n_points=2, batch_size=2, mc_points_size=64, num_mc_samples=256, mc_points_method="uniform", min_evals=100)
    # stage 1 (EI, one point per iteration) spends the whole budget: stage 2 must not evaluate anything
    b = BOBE(counted, ["x", "y"], bounds, n_sobol_init=8, seed=2, save=False)
    res = b.run(acq=("ei", "wipstd"), max_evals=12, **kw)
    assert res["n_evals"] == 12 and len(calls) == 12 and res["termination_reason"] == "Maximum evaluations reached"
    # stage 1 ends on ITS rule (a log-EI goal that the first iteration meets): stage 2 runs on with the rest of the budget
    calls.clear()
    d1 = str(tmp_path / "run")
    b = BOBE(counted, ["x", "y"], bounds, n_sobol_init=8, seed=2, likelihood_name="himmel", save=True, save_dir=d1, save_step=1)
    res = b.run(acq=("ei", "wipstd"), max_evals=15, ei_goal=1e300, **kw)
    assert res["gp"].npoints >= 13 and len(calls) == res["n_evals"]
    assert res["termination_reason"] == "Maximum evaluations reached"       # stage 2's own ending, not 'EI goal reached'
    st = json.load(open(tmp_path / "run" / "himmel_run.json"))
    assert st["stage"] == 1 and st["acq"].lower() == "wipstd"
    # resume: continues in stage 2 (no second EI stage), with a larger budget
    calls.clear()
    again = BOBE(counted, ["x", "y"], bounds, n_sobol_init=8, seed=9, likelihood_name="himmel", resume=True,
                 resume_file=str(tmp_path / "run" / "himmel"), save=False)
    assert not again.fresh_start and not calls
    n0 = again.gp.npoints
    res2 = again.run(acq=("ei", "wipstd"), max_evals=n0 + 4, ei_goal=1e300, **kw)
    assert res2["n_evals"] == n0 + 4 and len(calls) == 4 and again.acquisition.name.lower() == "wipstd"


def test_fit_of_a_surrogate_whose_hyperparameters_no_longer_factorise(caplog):
    """bo.py::_factorisable_start on the device: hyper-parameters at which K is numerically singular (``not_pd``: NaN factor,
    NaN predictions) - the fit starts from the incumbent with a smaller kernel variance instead of from the random starts
    alone, and ends with a factorised surrogate that still describes the data (the reference's fit would keep whichever
    uniform random start happens to be finite, optim.py:325-345)."""
    import logging
    from bobe_amd import GP
    from bobe_amd.bo import gp_fit
    rng = np.random.default_rng(8)
    n, d = 600, 5
    X = rng.uniform(size=(n, d))
    y = -np.sum((X - 0.4) ** 2, axis=1) - 0.3 * np.prod(X[:, :2], axis=1)
    gp = GP(X, y, noise=1e-8, lengthscales=np.full(d, 1.0), kernel_variance=10.0, pivot_floor_ulp=64.0)   # (as BOBE builds it)
    assert not gp.not_pd
    gp.update_hyperparams(np.log(np.append(np.full(d, 3.5), 1e7)))
    assert gp.not_pd and np.all(np.isnan(gp.predict_mean_batched(X[:3])))
    with caplog.at_level(logging.WARNING):
        res = gp_fit(gp, maxiters=40, n_restarts=4, rng=np.random.default_rng(1))
    assert any("no longer factorise" in r.getMessage() for r in caplog.records)
    assert np.isfinite(res["mll"]) and not gp.not_pd and gp.kernel_variance < 1e7
    assert np.min(gp.lengthscales) > 0.5                       # (not one of the random starts' short-length-scale optima)
    pred = gp.predict_mean_batched(X[:50])
    assert np.max(np.abs(pred - y[:50])) < 1e-3 * np.ptp(y)
