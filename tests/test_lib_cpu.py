"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/bobe_gp.h declares,
and fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bobe_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bobe_gp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bobe_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from bobe_amd import _lib
    declared = header_symbols()
    assert len(declared) >= 25
    bound = {name for name, _, _ in _lib.SIGNATURES}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bobe_gp.h but not exported"
        assert name in bound, f"{name} has no ctypes signature in bobe_amd/_lib.py"
    assert bound <= set(declared)
    assert b"gfx950" in lib.bobe_version()


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.bobe_device_count() == 0
    h = C.c_void_p(0)
    st = lib.bobe_gp_create(C.byref(h), 0, 0, 2)
    assert st < 0 and not h.value
    assert lib.bobe_last_error()
    from bobe_amd import GP, BobeLibraryError
    with pytest.raises(BobeLibraryError):
        GP(np.random.rand(5, 2), np.random.rand(5))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bobe_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{f} references the oracle"


def test_priors_match_oracle_and_gradients():
    from bobe_amd import priors as P
    from oracle import bobe_oracle as O
    x = np.array([0.05, 0.7, 3.0])
    pairs = [(P.Uniform(0.01, 5), lambda v: O._logpdf_uniform(v, 0.01, 5)),
             (P.LogNormal(0.3, 1.7), lambda v: O._logpdf_lognormal(v, 0.3, 1.7)),
             (P.HalfCauchy(0.1), lambda v: O._logpdf_halfcauchy(v, 0.1)),
             (P.Normal(0.2, 0.9), lambda v: O._logpdf_normal(v, 0.2, 0.9)),
             (P.HalfNormal(1.3), lambda v: O._logpdf_halfnormal(v, 1.3)),
             (P.Gamma(2.0, 3.0), lambda v: O._logpdf_gamma(v, 2.0, 3.0)),
             (P.dslp(4), lambda v: O._logpdf_lognormal(v, np.sqrt(2) + 0.5 * np.log(4), np.sqrt(3)))]
    for dist, ref in pairs:
        assert np.allclose(dist.log_prob(x), ref(x), rtol=1e-14, atol=1e-14)
        fd = (ref(x + 1e-6) - ref(x - 1e-6)) / 2e-6
        assert np.allclose(dist.dlog_prob(x), fd, rtol=1e-6, atol=1e-8)
    ls = np.array([0.2, 1.5, 0.7])
    lp, g_ls, g_kv, g_tau = P.saas_logprob_and_grad(ls, 2.0, 0.5)
    assert lp == pytest.approx(O.saas_prior_logprob(ls, 2.0, 0.5), rel=1e-14)
    e = 1e-6
    assert g_kv == pytest.approx((O.saas_prior_logprob(ls, 2.0 + e, 0.5) - O.saas_prior_logprob(ls, 2.0 - e, 0.5)) / (2 * e), rel=1e-6)
    assert g_tau == pytest.approx((O.saas_prior_logprob(ls, 2.0, 0.5 + e) - O.saas_prior_logprob(ls, 2.0, 0.5 - e)) / (2 * e), rel=1e-6)
    for j in range(3):
        lp_, lm_ = ls.copy(), ls.copy()
        lp_[j] += e
        lm_[j] -= e
        assert g_ls[j] == pytest.approx((O.saas_prior_logprob(lp_, 2.0, 0.5) - O.saas_prior_logprob(lm_, 2.0, 0.5)) / (2 * e), rel=1e-6)


def test_optimize_scipy_matches_oracle_driver():
    from bobe_amd.optim import optimize_scipy
    from oracle import bobe_oracle as O

    def vg(x):
        x = np.asarray(x)
        return float(np.sum((x - 0.3) ** 4) + np.sum(x ** 2)), 4 * (x - 0.3) ** 3 + 2 * x
    x0 = np.array([[0.9, -0.5], [0.1, 0.2], [2.0, 2.0]])
    a = optimize_scipy(vg, num_params=2, bounds=[-1, 3], x0=x0, maxiter=50, n_restarts=3, optimizer_options={})
    b = O.optimize_scipy(vg, 2, [-1, 3], x0, maxiter=50, n_restarts=3, optimizer_options={})
    assert np.allclose(a[0], b[0]) and a[1] == pytest.approx(b[1])
    with pytest.raises(ValueError):
        optimize_scipy(vg, num_params=2, bounds=[-1, 3], x0=x0[:1], n_restarts=2)


def test_synthetic_problem_is_seeded():
    from bobe_amd.synthetic import synthetic_problem, theta_schedule
    a = synthetic_problem(64, 3, 32, 16)
    b = synthetic_problem(64, 3, 32, 16)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    c = synthetic_problem(64, 3, 16, 16, cand_offset=16)
    assert np.array_equal(c[2], a[2][16:32])
    th = theta_schedule(3)
    assert th.shape == (20, 4) and th[0, 0] == pytest.approx(np.log(0.6)) and th[1, 0] == pytest.approx(np.log(0.6) - 0.05)


def test_compute_integrals_closed_form_and_resampling():
    """bobe_amd.samplers.compute_integrals restates the dynesty trapezoid of BOBE/samplers.py:27-50."""
    import math
    from bobe_amd.samplers import compute_integrals, renormalise_log_weights, resample_equal
    n, nlive = 4000, 100
    logvol = -np.arange(1, n + 1) / nlive
    z = compute_integrals(logl=np.zeros(n), logvol=logvol)
    assert np.all(np.diff(z) >= 0)
    # L = 1 with a zero-likelihood pad at X = 1: Z = (1 - X_n) - (1 - X_1)/2
    want = (1.0 - math.exp(logvol[-1])) - 0.5 * (1.0 - math.exp(logvol[0]))
    assert math.exp(z[-1]) == pytest.approx(want, rel=1e-12)
    z2 = compute_integrals(logl=np.zeros(n), logvol=logvol, squared=True)
    assert z2[-1] < z[-1]
    w = renormalise_log_weights(np.log(np.array([1.0, 3.0, 6.0])))
    assert np.allclose(w, [0.1, 0.3, 0.6])
    xs, ls = resample_equal(np.arange(3)[:, None], np.arange(3.0), w, rng=np.random.default_rng(0))
    assert xs.shape == (3, 1) and set(xs.ravel()) <= {0, 1, 2}


def test_concurrent_restart_drivers_equal_the_sequential_loop():
    """Host logic of the restart concurrency (no GPU): the slot driver (one thread per slot, restarts dealt round
    robin) and the lock-step batch driver must return exactly what the one-after-the-other loop returns."""
    from bobe_amd.optim import optimize_scipy

    def vg(x):
        x = np.asarray(x)
        f = float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    x0 = np.random.default_rng(0).uniform(-2, 2, size=(7, 4))
    seq = optimize_scipy(vg, num_params=4, bounds=[-3, 3], x0=x0, n_restarts=7)
    slot_calls, batch_sizes = [], []

    def slot_vg(x, slot):
        slot_calls.append(slot)
        return vg(x)

    def batch_vg(xs):
        batch_sizes.append(len(xs))
        return [vg(x) for x in xs]

    slots = optimize_scipy(vg, num_params=4, bounds=[-3, 3], x0=x0, n_restarts=7, slot_value_and_grad=slot_vg, n_slots=3)
    batch = optimize_scipy(vg, num_params=4, bounds=[-3, 3], x0=x0, n_restarts=7, batch_value_and_grad=batch_vg)
    assert np.array_equal(seq[0], slots[0]) and seq[1] == slots[1]
    assert np.array_equal(seq[0], batch[0]) and seq[1] == batch[1]
    assert set(slot_calls) == {0, 1, 2}                      # never more workers than slots
    assert max(batch_sizes) == 7 and min(batch_sizes) >= 1   # rounds shrink as restarts finish


def test_stepped_lbfgsb_driver_is_scipy_minimize_restart_for_restart():
    """The lock-step driver steps SciPy's reverse-communication L-BFGS-B routine from one thread (optim._rc_minimize_all,
    a restatement of scipy.optimize._lbfgsb_py._minimize_lbfgsb's loop on a private module): every restart must come
    out exactly as ``minimize`` returns it - iterate, value, iteration and evaluation counts, message - with bounds that
    bind, a maxiter that cuts some restarts short and an objective that goes NaN for one of them; and the thread driver
    (the fallback when the private routine is missing or different) must agree too."""
    from scipy.optimize import minimize
    from bobe_amd import optim

    def vg(x):
        x = np.asarray(x)
        f = float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2) + 0.01 * np.sum(np.cos(5 * x)))
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        g += -0.05 * np.sin(5 * x)
        if x[0] > 1.9:                                        # (like a kernel matrix that is not positive definite)
            return float("nan"), np.full_like(x, np.nan)
        return f, g

    assert optim._rc_available()                              # this image's SciPy has the routine the loop was written against
    starts = np.random.default_rng(3).uniform(-2, 2, size=(6, 5))
    starts[2, 0] = 1.95
    bounds = [(-2.0, 2.0), (-0.5, 2.0), (None, 1.5), (-2.0, None), (None, None)]
    for maxiter, tol in ((5, {"ftol": 1e-9, "gtol": 1e-7}), (200, {"ftol": 1e-9, "gtol": 1e-7}), (200, {"ftol": 1e-15, "gtol": 1e-12})):
        # (the last setting drives the line search into failures: the routine then asks again at the point it has)
        kw = dict(method="L-BFGS-B", bounds=bounds, options=dict(tol, maxiter=maxiter))
        ref = [minimize(vg, x0, jac=True, **kw) for x0 in starts]
        sizes = []

        def batch(xs):
            sizes.append(len(xs))
            return [vg(x) for x in xs]
        got = optim._minimize_concurrently(batch, starts, True, **kw)
        assert max(sizes) == 6 and min(sizes) >= 1
        saved = optim._RC_STATE["ok"]
        optim._RC_STATE["ok"] = False                         # the thread driver
        try:
            thr = optim._minimize_concurrently(lambda xs: [vg(x) for x in xs], starts, True, **kw)
        finally:
            optim._RC_STATE["ok"] = saved
        for r, a, b in zip(ref, got, thr):
            for o in (a, b):
                assert np.array_equal(o.x, r.x, equal_nan=True) and (o.fun == r.fun or (np.isnan(o.fun) and np.isnan(r.fun)))
                assert o.nit == r.nit and o.nfev == r.nfev and o.message == r.message and bool(o.success) == bool(r.success)


def test_small_host_helpers():
    from bobe_amd.dist_sweep import dist_info, shard_bounds
    from bobe_amd.samplers import get_hmc_settings, prior_transform
    from bobe_amd.utils import get_threshold_for_nsigma
    assert dist_info() == (1, 0, None)                       # no process group: single rank
    assert [shard_bounds(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]     # np.array_split boundaries
    assert get_hmc_settings(4) == (256, 1024, 4) and get_hmc_settings(10, thinning=2) == (512, 2048, 2)
    assert prior_transform(0.3) == 0.3
    # utils/core.py:150-167: 1 sigma in 1-D is half a chi-square unit; grows with the dimension
    assert get_threshold_for_nsigma(1.0, 1) == pytest.approx(0.5, rel=1e-9)
    assert get_threshold_for_nsigma(2.0, 2) > get_threshold_for_nsigma(2.0, 1) > get_threshold_for_nsigma(1.0, 1)


def _build_c_host(tmp_path):
    """examples/c_abi_host.c: a host program in plain C99 over include/bobe_gp.h (no Python, no torch)"""
    import subprocess
    exe = str(tmp_path / "c_abi_host")
    lib_dir = os.path.join(ROOT, "bobe_amd")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_host.c"), "-o", exe, "-L", lib_dir, "-lbobe_gp", f"-Wl,-rpath,{lib_dir}", "-lm"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_header_is_plain_c_and_a_c_host_links_against_the_library(tmp_path, lib):
    """The drop-in boundary is a C ABI: the header must compile as strict C99 and a C program must link against
    libbobe_gp.so without any C++ / HIP / Python on its side.  Without a GPU it reports that and exits 77 (no CPU path)."""
    import subprocess
    exe = _build_c_host(tmp_path)
    if lib.bobe_device_count() >= 1:
        pytest.skip("a GPU is present: tests/test_gpu_parity.py runs the program")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 77 and "no CPU compute path" in p.stderr


def test_an_exception_in_one_restart_ends_that_restart_only():
    """an exception out of the batch objective for one restart's point used to abort every restart and run
    the whole fit again under the thread driver.  Now the round's points are evaluated one by one, the offending
    restart ends with its exception (optimize_scipy skips it like any failed restart) and the others finish as
    ``minimize`` would - under the stepped driver and under the thread driver alike."""
    from scipy.optimize import minimize
    from bobe_amd import optim

    def vg(x):
        x = np.asarray(x)
        if x[0] < -1.5:
            raise FloatingPointError("objective undefined here")
        return float(np.sum((x - 0.3) ** 2) + 0.1 * np.sum(x ** 4)), 2 * (x - 0.3) + 0.4 * x ** 3

    starts = np.array([[1.0, -1.0, 0.5], [-1.8, 0.2, 0.1], [0.9, 0.9, -0.9]])
    kw = dict(method="L-BFGS-B", bounds=[(-2, 2)] * 3, options={"maxiter": 50})
    ref = [minimize(vg, starts[i], jac=True, **kw) for i in (0, 2)]
    calls = []

    def batch(xs):
        calls.append(len(xs))
        return [vg(x) for x in xs]
    for force_threads in (False, True):
        saved = optim._RC_STATE["ok"]
        assert optim._rc_available()
        if force_threads:
            optim._RC_STATE["ok"] = False
        try:
            out = optim._minimize_concurrently(batch, starts, True, **kw)
        finally:
            optim._RC_STATE["ok"] = saved
        assert isinstance(out[1], FloatingPointError)
        for r, o in zip(ref, (out[0], out[2])):
            assert np.array_equal(o.x, r.x) and o.fun == r.fun and o.nit == r.nit and o.nfev == r.nfev
    best_x, best_f = optim.optimize_scipy(vg, num_params=3, bounds=[-2, 2], x0=starts, n_restarts=3, batch_value_and_grad=batch)
    # (optimize_scipy's own options differ from `kw`: same optimum, not the same last digits)
    assert np.allclose(best_x, ref[0].x, atol=1e-3) and best_f == pytest.approx(min(r.fun for r in ref), rel=1e-4)
    assert optim.lbfgs_driver() == "stepped"


def test_stepped_driver_is_gated_on_the_routine_not_on_the_version_text(monkeypatch):
    """The stepped L-BFGS-B driver is tied to SciPy's private ``setulb``: what it checks is the routine's own argument list
    (its docstring) and the message tables, not ``scipy.__version__``.  Another version string with the same routine keeps the
    stepped driver; a routine with another argument list (or none stated) switches to the thread driver - and that
    fallback gives ``minimize``'s iterates, bit for bit."""
    import scipy
    from scipy.optimize import _lbfgsb, minimize
    from bobe_amd import optim

    def fresh():
        optim._RC_STATE.update(checked=False, ok=False)
        return optim._rc_available()

    saved = dict(optim._RC_STATE)
    try:
        assert optim._setulb_mismatch() is None and fresh()
        monkeypatch.setattr(scipy, "__version__", "9.99.0")                      # the version text alone changes nothing
        assert optim._setulb_mismatch() is None and fresh() and optim.lbfgs_driver() == "stepped"

        class Other:                                                             # a routine with one argument more
            __doc__ = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,iprint,lsave,isave,dsave,maxls,ln_task)"

            def __call__(self, *a):
                raise AssertionError("the stepped driver must not call a routine it was not written for")
        monkeypatch.setattr(_lbfgsb, "setulb", Other())
        assert "setulb takes" in optim._setulb_mismatch() and not fresh() and optim.lbfgs_driver() == "threads"

        def vg(x):
            x = np.asarray(x)
            return float(np.sum((x - 0.3) ** 2) + 0.1 * np.sum(x ** 4)), 2 * (x - 0.3) + 0.4 * x ** 3
        starts = np.array([[1.0, -1.0, 0.5], [-1.2, 0.2, 0.1], [0.9, 0.9, -0.9]])
        kw = dict(method="L-BFGS-B", bounds=[(-2, 2)] * 3, options={"maxiter": 50})
        monkeypatch.undo()                                                       # (minimize itself needs the real routine;
        optim._RC_STATE.update(checked=True, ok=False)                           #  the driver decision stays "threads")
        ref = [minimize(vg, x0, jac=True, **kw) for x0 in starts]
        out = optim._minimize_concurrently(lambda xs: [vg(x) for x in xs], starts, True, **kw)
        for r, o in zip(ref, out):
            assert np.array_equal(o.x, r.x) and o.fun == r.fun and o.nit == r.nit and o.nfev == r.nfev and o.message == r.message
        for doc in (None, "", "reverse communication routine"):                  # no argument list stated: switched off
            class NoDoc:
                __doc__ = doc
            monkeypatch.setattr(_lbfgsb, "setulb", NoDoc())
            assert optim._setulb_mismatch() is not None
            monkeypatch.undo()
    finally:
        optim._RC_STATE.update(saved)


def test_fit_start_is_walked_back_to_a_factorisable_kernel_variance():
    """bo.py::_factorisable_start: when the surrogate's hyper-parameters no longer factorise (``not_pd``), row 0 of the
    fit's starts is the incumbent with its kernel variance lowered by factors of four until the objective is finite;
    otherwise (factorised state, fixed kernel variance, nothing finite down to the bound) the incumbent itself."""
    import math
    import numpy as np
    from bobe_amd.bo import _factorisable_start

    class Fake:
        ndim, npoints, fixed_kernel_variance, not_pd = 3, 1000, False, True
        hyperparam_bounds = np.log(np.array([[0.01] * 3 + [1e-4], [5.0] * 3 + [1e8]]))
        wall = math.log(3.0e4)
        calls = 0

        def neg_mll_value_and_grad_batch(self, thetas, want_grad=True):
            Fake.calls += 1
            assert not want_grad and len(thetas) <= 8
            return [((float("nan") if t[3] > self.wall else -100.0 - t[3]), None) for t in thetas]

    init = np.log(np.array([2.0, 3.0, 4.0, 1.0e6]))
    out = _factorisable_start(Fake(), init)
    assert np.array_equal(out[:3], init[:3]) and out[3] <= Fake.wall < out[3] + math.log(4.0) and Fake.calls == 1
    assert math.isclose(math.exp(out[3]), 1.0e6 / 4 ** 3)                       # 2.5e5, 6.25e4 are above the wall; 1.5625e4 is not
    ok = Fake()
    ok.not_pd = False
    assert _factorisable_start(ok, init) is init
    fixed = Fake()
    fixed.fixed_kernel_variance = True
    assert _factorisable_start(fixed, init) is init
    hopeless = Fake()
    hopeless.wall = -math.inf
    Fake.calls = 0
    assert _factorisable_start(hopeless, init) is init and Fake.calls == 2     # sixteen candidates, eight per batch


def test_import_BOBE_alias_resolves_to_bobe_amd():
    """bobe_amd/compat on the path: ``import BOBE`` and the reference's submodule paths bind to this package (no GPU is
    touched by importing)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import BOBE, bobe_amd\n"
            "from BOBE import BOBE as B, GP, Likelihood, WIPStd, scale_from_unit, get_logger\n"
            "from BOBE.gp import GP as G2, rbf_kernel, gp_mll\n"
            "from BOBE.bo import BOBE as B2, load_gp_file\n"
            "from BOBE.utils.core import scale_to_unit, renormalise_log_weights, resample_equal\n"
            "from BOBE.utils.seed import set_global_seed, get_numpy_rng\n"
            "from BOBE.utils import get_logger as gl2\n"
            "from BOBE.acquisition import EI, LogEI, WIPV, get_mc_samples\n"
            "from BOBE.samplers import nested_sampling_Dy, sample_GP_NUTS\n"
            "from BOBE.optim import optimize_scipy\n"
            "from BOBE.clf import svm_predict, CLASSIFIER_REGISTRY\n"
            "from BOBE.clf_gp import GPwithClassifier\n"
            "assert B is bobe_amd.BOBE is B2 and GP is bobe_amd.GP is G2 and BOBE.gp is bobe_amd.gp\n"
            "try:\n    BOBE.BOBEResults\n    raise SystemExit('results manager should not exist')\n"
            "except AttributeError as e:\n    assert 'does not build' in str(e)\n"
            "print('alias ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "bobe_amd", "compat"), root,
                                                        os.environ.get("PYTHONPATH", "")]))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "alias ok" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
