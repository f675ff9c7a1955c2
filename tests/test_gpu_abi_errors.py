"""The boundary's error contract (include/bobe_gp.h, SURVEY 8b "Errors"): a misused entry point returns a negative status
and leaves its text in bobe_last_error(); nothing throws, aborts or faults across the C ABI, and the handle stays usable
afterwards.  (Numerical conditions are the positive statuses: BOBE_NOT_PD with NaN outputs, like XLA's NaN factor in the
reference, optim.py:328, 341 - covered in test_gpu_parity.py.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_HIP, ERR_STATE = -1, -2, -3


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def test_every_entry_point_refuses_a_null_handle_and_the_wrong_state():
    from bobe_amd import _lib
    lib = _lib.load()
    d, n = 3, 40
    rng = np.random.default_rng(0)
    X, y = np.ascontiguousarray(rng.uniform(size=(n, d))), np.ascontiguousarray(rng.normal(size=n))
    ls = np.full(d, 0.5)
    q, Z = np.ascontiguousarray(rng.uniform(size=(8, d))), np.ascontiguousarray(rng.uniform(size=(16, d)))
    o8, o8b, g8, g8b = np.empty(8), np.empty(8), np.empty((8, d)), np.empty((8, d))
    val, i64a, i64b, f64a, f64b = C.c_double(), C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
    acc, ins = np.zeros(8, np.int32), np.zeros(8, np.int32)
    state, adapt = np.zeros((8, 3 * d + 2)), np.tile(np.array([0.1, 0.0, 0.0, 0.0, 0.0]), (8, 1))
    L, al = np.empty((n, n)), np.empty(n)

    def calls(h):
        """Every entry point that takes a handle, with otherwise valid arguments."""
        return {
            "set_stream": lambda: lib.bobe_gp_set_stream(h, None),
            "sync": lambda: lib.bobe_gp_sync(h),
            "set_chunk": lambda: lib.bobe_gp_set_chunk(h, 8192),
            "set_data": lambda: lib.bobe_gp_set_data(h, _p(X), _p(y), n),
            "set_hyper": lambda: lib.bobe_gp_set_hyper(h, _p(ls), 1.0, 1e-6),
            "factor": lambda: lib.bobe_gp_factor(h),
            "mll": lambda: lib.bobe_gp_mll(h, _p(ls), 1.0, C.byref(val), None),
            "mll_batch": lambda: lib.bobe_gp_mll_batch(h, 1, _p(ls), _p(np.ones(1)), _p(o8), None, None),
            "mll_submit": lambda: lib.bobe_gp_mll_submit(h, 0, _p(ls), 1.0, 1),
            "predict": lambda: lib.bobe_gp_predict(h, _p(q), 8, _p(o8), _p(o8b), 1),
            "wip_sweep": lambda: lib.bobe_gp_wip_sweep(h, _p(q), 8, _p(Z), 16, 1.0, _p(o8), _p(o8b), None, None,
                                                       C.byref(i64a), C.byref(f64a), C.byref(i64b), C.byref(f64b)),
            "fantasy_var": lambda: lib.bobe_gp_fantasy_var(h, _p(q), 8, _p(Z), 16, 1.0, _p(np.empty((8, 16)))),
            "wip_grad": lambda: lib.bobe_gp_wip_grad(h, _p(q), 8, _p(Z), 16, 1.0, _p(o8), _p(o8b), _p(g8), _p(g8b)),
            "acq_ei": lambda: lib.bobe_gp_acq_ei(h, _p(q), 8, 0.0, 0.0, 0, _p(o8)),
            "predict_grad": lambda: lib.bobe_gp_predict_grad(h, _p(q), 8, _p(o8), _p(o8b), _p(g8), _p(g8b)),
            "hmc_leapfrog": lambda: lib.bobe_gp_hmc_leapfrog(h, 8, _p(g8.copy()), _p(g8b.copy()), _p(np.ones(d)), 0.1, 2, 1.0,
                                                             0.0, 1.0, _p(o8), _p(g8), _p(o8b), _p(g8b)),
            "hmc_run": lambda: lib.bobe_gp_hmc_run(h, 8, _p(state), _p(adapt), _p(np.ones(d)), 1, 0, 2, 1, 1.0, 0.0, 1.0, 0,
                                                   None, 1, None, None),
            "rwalk": lambda: lib.bobe_gp_rwalk(h, 8, _p(q.copy()), _p(o8), _p(0.1 * np.eye(d)), -1e30, 2, 1, 1.0, 0.0,
                                               _p(acc), _p(ins), None),
            "gate_eval": lambda: lib.bobe_gp_gate_eval(h, _p(q), 8, _p(o8), _p(o8b)),
            "get_chol": lambda: lib.bobe_gp_get_chol(h, _p(L), _p(al)),
            "append": lambda: lib.bobe_gp_append(h, _p(q[:1]), 1, _p(np.zeros(n + 1))),
            "debug_kinv": lambda: lib.bobe_debug_kinv(h, _p(L)),
            "debug_linv": lambda: lib.bobe_debug_linv(h, _p(L)),
            "time_potrf": lambda: lib.bobe_debug_time_potrf(h, 1, C.byref(val)),
        }

    # (i) NULL handle: ERR_ARG and a message from every one of them (bobe_gp_destroy(NULL) is a no-op like free)
    for name, call in calls(None).items():
        rc = call()
        assert rc == ERR_ARG and lib.bobe_last_error(), (name, rc)
    lib.bobe_gp_destroy(None)
    assert lib.bobe_gp_mll_wait(None, 0, C.byref(val), None) == ERR_ARG
    assert lib.bobe_gp_set_gate(None, None, 0, None, 0.0, 1.0, 0.5, -1e5) == ERR_ARG
    assert lib.bobe_gp_clone_state(None, None) < 0

    # (ii) creation: bad dimension / kernel id / device / NULL out
    h = C.c_void_p(0)
    assert lib.bobe_gp_create(None, 0, 0, d) == ERR_ARG
    for dim in (0, -1, 33):
        assert lib.bobe_gp_create(C.byref(h), 0, 0, dim) == ERR_ARG and not h.value
    assert lib.bobe_gp_create(C.byref(h), 0, 7, d) == ERR_ARG and not h.value
    assert lib.bobe_gp_create(C.byref(h), 99, 0, d) == ERR_HIP and b"no such HIP device" in lib.bobe_last_error()
    assert lib.bobe_gp_create(C.byref(h), 0, 0, d) == 0 and h.value

    # (iii) a fresh handle holds no data: everything that needs data or a factor says so (ERR_STATE), the rest works
    no_state_needed = {"set_stream", "sync", "set_chunk", "set_data", "set_hyper"}
    stream_before = lib.bobe_gp_get_stream(h)
    for name, call in calls(h).items():
        if name in no_state_needed:
            continue
        rc = call()
        assert rc == ERR_STATE and lib.bobe_last_error(), (name, rc, lib.bobe_last_error())
    assert lib.bobe_gp_mll_wait(h, 0, C.byref(val), None) == ERR_STATE            # nothing submitted
    assert lib.bobe_gp_get_stream(h) == stream_before

    # (iv) data but no factor: the evaluations work, the consumers of the factor still refuse
    assert lib.bobe_gp_set_data(h, _p(X), _p(y), n) == 0
    assert lib.bobe_gp_set_hyper(h, _p(ls), 1.0, 1e-6) == 0
    needs_factor = {"predict", "wip_sweep", "fantasy_var", "wip_grad", "acq_ei", "predict_grad", "hmc_leapfrog", "hmc_run",
                    "rwalk", "get_chol", "append", "debug_kinv", "debug_linv"}
    c = calls(h)
    for name in sorted(needs_factor):
        rc = c[name]()
        assert rc == ERR_STATE, (name, rc, lib.bobe_last_error())
    assert c["mll"]() == 0 and np.isfinite(val.value)
    assert c["gate_eval"]() == ERR_STATE and b"no classifier gate" in lib.bobe_last_error()

    # (v) factorised: bad sizes and missing arrays are ERR_ARG; the handle keeps working after every refusal
    assert lib.bobe_gp_factor(h) == 0
    assert c["predict"]() == 0
    ref = o8.copy()
    bad = {
        "predict C=0": lambda: lib.bobe_gp_predict(h, _p(q), 0, _p(o8), None, 1),
        "predict C<0": lambda: lib.bobe_gp_predict(h, _p(q), -5, _p(o8), None, 1),
        "predict Xq NULL": lambda: lib.bobe_gp_predict(h, None, 8, _p(o8), None, 1),
        "sweep M=0": lambda: lib.bobe_gp_wip_sweep(h, _p(q), 8, _p(Z), 0, 1.0, _p(o8b), None, None, None, None, None, None, None),
        "sweep Z NULL": lambda: lib.bobe_gp_wip_sweep(h, _p(q), 8, None, 16, 1.0, _p(o8b), None, None, None, None, None, None, None),
        "wip_grad C=0": lambda: lib.bobe_gp_wip_grad(h, _p(q), 0, _p(Z), 16, 1.0, _p(o8b), None, None, None),
        "acq_ei out NULL": lambda: lib.bobe_gp_acq_ei(h, _p(q), 8, 0.0, 0.0, 0, None),
        "predict_grad var without dvar": lambda: lib.bobe_gp_predict_grad(h, _p(q), 8, _p(o8b), _p(o8b), _p(g8), None),
        "hmc_leapfrog L=0": lambda: lib.bobe_gp_hmc_leapfrog(h, 8, _p(g8.copy()), _p(g8b.copy()), _p(np.ones(d)), 0.1, 0, 1.0,
                                                             0.0, 1.0, _p(o8b), _p(g8), _p(o8b), _p(g8b)),
        "hmc_leapfrog temp=0": lambda: lib.bobe_gp_hmc_leapfrog(h, 8, _p(g8.copy()), _p(g8b.copy()), _p(np.ones(d)), 0.1, 2,
                                                                1.0, 0.0, 0.0, _p(o8b), _p(g8), _p(o8b), _p(g8b)),
        "hmc_run P=0": lambda: lib.bobe_gp_hmc_run(h, 0, _p(state), _p(adapt), _p(np.ones(d)), 1, 0, 2, 1, 1.0, 0.0, 1.0, 0, None,
                                                   1, None, None),
        "rwalk walks=0": lambda: lib.bobe_gp_rwalk(h, 8, _p(q.copy()), _p(o8b), _p(0.1 * np.eye(d)), -1e30, 0, 1, 1.0, 0.0,
                                                   _p(acc), _p(ins), None),
        "rwalk counters NULL": lambda: lib.bobe_gp_rwalk(h, 8, _p(q.copy()), _p(o8b), _p(0.1 * np.eye(d)), -1e30, 2, 1, 1.0, 0.0,
                                                         None, None, None),
        "set_data N=0": lambda: lib.bobe_gp_set_data(h, _p(X), _p(y), 0),
        "set_chunk 100": lambda: lib.bobe_gp_set_chunk(h, 100),
        "mll_submit slot 8": lambda: lib.bobe_gp_mll_submit(h, 8, _p(ls), 1.0, 1),
        "mll_submit slot -1": lambda: lib.bobe_gp_mll_submit(h, -1, _p(ls), 1.0, 1),
        "mll_batch B<0": lambda: lib.bobe_gp_mll_batch(h, -1, _p(ls), _p(np.ones(1)), _p(o8b), None, None),
        "set_gate n_sv<0": lambda: lib.bobe_gp_set_gate(h, _p(q), -1, _p(o8b), 0.0, 1.0, 0.5, -1e5),
        "kernel include_noise on a rectangle": lambda: lib.bobe_gp_kernel(h, _p(q), 8, _p(Z), 16, _p(ls), 1.0, 1e-6, 1,
                                                                          _p(np.empty((8, 16)))),
        "kernel empty": lambda: lib.bobe_gp_kernel(h, _p(q), 0, _p(Z), 16, _p(ls), 1.0, 1e-6, 0, _p(np.empty((8, 16)))),
    }
    for name, call in bad.items():
        rc = call()
        assert rc == ERR_ARG and lib.bobe_last_error(), (name, rc, lib.bobe_last_error())
        assert c["predict"]() == 0 and np.array_equal(o8, ref), name               # same handle, same bits afterwards
    # a slot can hold one evaluation; waiting twice is a state error, not a hang
    assert lib.bobe_gp_mll_submit(h, 2, _p(ls), 1.0, 1) == 0
    assert lib.bobe_gp_mll_submit(h, 2, _p(ls), 1.0, 1) == ERR_STATE
    gr = np.empty(d + 1)
    assert lib.bobe_gp_mll_wait(h, 2, C.byref(val), _p(gr)) == 0 and np.isfinite(val.value) and np.all(np.isfinite(gr))
    assert lib.bobe_gp_mll_wait(h, 2, C.byref(val), _p(gr)) == ERR_STATE
    # an empty lock-step batch is a no-op
    assert lib.bobe_gp_mll_batch(h, 0, _p(ls), _p(np.ones(1)), _p(o8b), None, None) == 0
    # clone between unlike handles
    h2 = C.c_void_p(0)
    assert lib.bobe_gp_create(C.byref(h2), 0, 1, d) == 0
    assert lib.bobe_gp_clone_state(h2, h) == ERR_ARG and b"same kernel" in lib.bobe_last_error()
    lib.bobe_gp_destroy(h2)
    assert c["predict"]() == 0 and np.array_equal(o8, ref)
    lib.bobe_gp_destroy(h)


def test_round5_entry_points_error_contract():
    """bobe_gp_dist_sq / _mll_from_k / _chol_row_update / _set_pivot_floor_ulp / _set_refine_kappa / _get_refine: NULL and
    bad sizes are ERR_ARG with a message, a handle that holds training data of another size refuses a free-function call
    with ERR_STATE (it would have to resize the workspace under the data), and a refused call leaves the handle as it was."""
    from bobe_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    d, n = 2, 30
    X, y = np.ascontiguousarray(rng.uniform(size=(n, d))), np.ascontiguousarray(rng.normal(size=n))
    K = np.ascontiguousarray(np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.25) + 1e-4 * np.eye(n))
    L = np.ascontiguousarray(np.linalg.cholesky(K))
    out, v, val, dg = np.empty((n, n)), np.empty(n), C.c_double(), C.c_double()
    kap, act = C.c_double(), C.c_int()
    # NULL handle
    assert lib.bobe_gp_dist_sq(None, _p(X), n, _p(X), n, _p(out)) == ERR_ARG and lib.bobe_last_error()
    assert lib.bobe_gp_mll_from_k(None, _p(K), n, _p(y), C.byref(val)) == ERR_ARG
    assert lib.bobe_gp_chol_row_update(None, _p(L), n, _p(y), 2.0, _p(v), C.byref(dg)) == ERR_ARG
    assert lib.bobe_gp_set_pivot_floor_ulp(None, 64.0) == ERR_ARG and lib.bobe_gp_get_pivot_floor_ulp(None) == -1.0
    assert lib.bobe_gp_set_refine_kappa(None, 1e6) == ERR_ARG and lib.bobe_gp_get_refine(None, C.byref(kap), C.byref(act)) == ERR_ARG
    h = C.c_void_p(0)
    assert lib.bobe_gp_create(C.byref(h), 0, 0, d) == 0
    try:
        # data-less handle: the free functions work and leave it data-less; NULL arrays / n < 1 are refused
        assert lib.bobe_gp_dist_sq(h, _p(X), n, _p(X), n, _p(out)) == 0 and np.allclose(np.diag(out), 0.0)
        assert lib.bobe_gp_mll_from_k(h, _p(K), n, _p(y), C.byref(val)) == 0 and np.isfinite(val.value)
        assert lib.bobe_gp_chol_row_update(h, _p(L), n, _p(K[:, 0].copy()), 5.0, _p(v), C.byref(dg)) == 0 and np.isfinite(dg.value)
        assert lib.bobe_gp_npoints(h) in (0, n) and lib.bobe_gp_factor(h) == ERR_STATE          # still no training data
        for rc in (lib.bobe_gp_dist_sq(h, None, n, _p(X), n, _p(out)), lib.bobe_gp_dist_sq(h, _p(X), 0, _p(X), n, _p(out)),
                   lib.bobe_gp_mll_from_k(h, _p(K), 0, _p(y), C.byref(val)), lib.bobe_gp_mll_from_k(h, None, n, _p(y), C.byref(val)),
                   lib.bobe_gp_mll_from_k(h, _p(K), n, _p(y), None), lib.bobe_gp_chol_row_update(h, _p(L), -3, _p(y), 1.0, _p(v), C.byref(dg)),
                   lib.bobe_gp_chol_row_update(h, _p(L), n, None, 1.0, _p(v), C.byref(dg)),
                   lib.bobe_gp_set_pivot_floor_ulp(h, -1.0), lib.bobe_gp_set_pivot_floor_ulp(h, float("nan")),
                   lib.bobe_gp_set_refine_kappa(h, float("nan"))):
            assert rc == ERR_ARG and lib.bobe_last_error(), rc
        assert lib.bobe_gp_get_pivot_floor_ulp(h) == 0.0                       # the reference's sign-only rule
        assert lib.bobe_gp_get_solve_block(h) == 128 and lib.bobe_gp_get_solve_block(None) == -1
        for rc in (lib.bobe_gp_set_solve_block(None, 128), lib.bobe_gp_set_solve_block(h, 0), lib.bobe_gp_set_solve_block(h, 100),
                   lib.bobe_debug_solve_opts(h, 100, 0), lib.bobe_debug_solve_opts(h, 512, 77), lib.bobe_debug_solve_opts(None, 512, 0)):
            assert rc == ERR_ARG and lib.bobe_last_error(), rc
        assert lib.bobe_gp_set_solve_block(h, 256) == 0 and lib.bobe_gp_get_solve_block(h) == 256
        assert lib.bobe_gp_get_refine(h, C.byref(kap), C.byref(act)) == 0 and kap.value == 1e6 and act.value == 0
        assert lib.bobe_gp_get_refine(h, None, None) == 0
        # a handle with training data of ANOTHER size refuses the free functions on matrices (ERR_STATE) and stays intact
        ls = np.full(d, 0.5)
        assert lib.bobe_gp_set_data(h, _p(X[:20].copy()), _p(y[:20].copy()), 20) == 0
        assert lib.bobe_gp_set_hyper(h, _p(ls), 1.0, 1e-6) == 0 and lib.bobe_gp_factor(h) == 0
        m0 = np.empty(4)
        assert lib.bobe_gp_predict(h, _p(X[:4].copy()), 4, _p(m0), None, 1) == 0
        assert lib.bobe_gp_mll_from_k(h, _p(K), n, _p(y), C.byref(val)) == ERR_STATE and b"another size" in lib.bobe_last_error()
        assert lib.bobe_gp_chol_row_update(h, _p(L), n, _p(y), 2.0, _p(v), C.byref(dg)) == ERR_STATE
        m1 = np.empty(4)
        assert lib.bobe_gp_predict(h, _p(X[:4].copy()), 4, _p(m1), None, 1) == 0 and np.array_equal(m0, m1)
        assert lib.bobe_gp_dist_sq(h, _p(X), n, _p(X), n, _p(out)) == 0                          # (needs no workspace of the factor)
    finally:
        lib.bobe_gp_destroy(h)
