"""Stage-by-stage GPU-vs-oracle diagnostics (prints error metrics; asserts nothing).
Run on the GPU box:  python tools/gpu_diag.py [N d]  > gpurun_out/diag.log"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402
from oracle import bobe_oracle as O  # noqa: E402
from scipy.linalg import solve_triangular  # noqa: E402

lib = _lib.load()
print(lib.bobe_version().decode(), "devices:", lib.bobe_device_count(), flush=True)


def gemm_check():
    rng = np.random.default_rng(0)
    M, N, K = 256, 384, 160
    for la in (0, 1):
        for lb in (0, 1):
            A = rng.standard_normal((M, K))
            B = rng.standard_normal((N, K))
            Aarr = np.ascontiguousarray(A if la == 0 else A.T)
            Barr = np.ascontiguousarray(B if lb == 0 else B.T)
            Cc = np.zeros((M, N))
            st = lib.bobe_debug_gemm(0, la, lb, M, N, K, _lib.ptr(Aarr), Aarr.shape[1], _lib.ptr(Barr), Barr.shape[1],
                                     _lib.ptr(Cc), N)
            ref = A @ B.T
            print(f"gemm la={la} lb={lb} status={st} maxerr={np.max(np.abs(Cc - ref)):.3e} "
                  f"(ref scale {np.max(np.abs(ref)):.2f}) {lib.bobe_last_error().decode() if st else ''}", flush=True)


def stage_check(n, d, kernel="rbf", ls0=0.5, C=700, M=100):
    rng = np.random.default_rng(1)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + np.cos(2 * X[:, -1]) + 0.1 * rng.standard_normal(n)
    ls = np.full(d, ls0) * (1 + 0.1 * np.arange(d))
    kvar, noise = 1.3, 1e-6
    t0 = time.time()
    gp = GP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    print(f"--- N={n} d={d} kernel={kernel}: GP built+factored in {time.time() - t0:.3f}s not_pd={gp.not_pd}", flush=True)
    og = O.OracleGP(X, y, noise=noise, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    Kd = gp.kernel(X, X, include_noise=True)
    Ko = og.kernel(X, X, ls, kvar, noise, include_noise=True)
    print(f"K     maxabs err {np.max(np.abs(Kd - Ko)):.3e}")
    L = gp.cholesky
    print(f"L     maxabs err {np.max(np.abs(L - og.cholesky)):.3e}  upper zero: {np.max(np.abs(np.triu(L, 1))):.1e}")
    print(f"LLt-K maxabs     {np.max(np.abs(L @ L.T - Ko)):.3e}")
    Li = np.empty((n, n))
    lib.bobe_debug_linv(gp._h, _lib.ptr(Li))
    Lio = solve_triangular(og.cholesky, np.eye(n), lower=True)
    print(f"Linv  maxabs err {np.max(np.abs(Li - Lio)):.3e} (scale {np.max(np.abs(Lio)):.2e})  |Linv L - I| {np.max(np.abs(Li @ og.cholesky - np.eye(n))):.3e}")
    a = gp.alphas.ravel()
    print(f"alpha rel err    {np.max(np.abs(a - og.alphas.ravel())) / np.max(np.abs(og.alphas)):.3e}")
    Ki = np.empty((n, n))
    lib.bobe_debug_kinv(gp._h, _lib.ptr(Ki))
    Kio = Lio.T @ Lio
    print(f"Kinv  rel err    {np.max(np.abs(Ki - Kio)) / np.max(np.abs(Kio)):.3e}")
    th = np.log(np.append(ls, kvar)) + 0.03
    f, g = gp.neg_mll_value_and_grad(th)
    fo, go = og.neg_mll_value_and_grad(th)
    print(f"mll   {f:.12e} vs {fo:.12e} rel {abs(f - fo) / abs(fo):.3e}")
    print(f"grad  rel err    {np.max(np.abs(g - go)) / np.max(np.abs(go)):.3e}\n   gpu {g}\n   cpu {go}")
    cand = rng.uniform(size=(C, d))
    cand[0] = X[3]
    Z = rng.uniform(size=(M, d))
    r = gp.wip_sweep(cand, Z, want_mean_var=True)
    ro = O.wip_sweep(og, cand, Z)
    for k in ("mean", "var", "wipv", "wipstd"):
        print(f"sweep {k:7s} maxabs err {np.max(np.abs(r[k] - ro[k])):.3e} (scale {np.max(np.abs(ro[k])):.2e})")
    print(f"argmin v {r['argmin_v']} vs {ro['argmin_v']}  s {r['argmin_s']} vs {ro['argmin_s']}  min_s {r['min_s']:.6e} vs {ro['wipstd'].min():.6e}")
    fv = gp.fantasy_var(cand[:5], Z)
    fo_ = np.array([og.fantasy_var(c, Z, og._k12(Z)) for c in cand[:5]])
    print(f"fantasy_var maxabs err {np.max(np.abs(fv - fo_)):.3e}")
    pm = gp.predict_mean_batched(cand[:50])
    pv = gp.predict_var_batched(cand[:50])
    print(f"predict mean err {np.max(np.abs(pm - og.predict_mean_batched(cand[:50]))):.3e} var err {np.max(np.abs(pv - og.predict_var_batched(cand[:50]))):.3e}")
    best = float(np.max(og.train_y))
    m_, v_ = og.predict_batched(cand[:50])
    print(f"EI err {np.max(np.abs(gp.acq_ei(cand[:50], best) - O.ei_score(m_, v_, best))):.3e} "
          f"logEI err {np.max(np.abs(gp.acq_ei(cand[:50], best, log_ei=True) - O.log_ei_score(m_, v_, best))):.3e}", flush=True)


if __name__ == "__main__":
    gemm_check()
    stage_check(100, 2)
    stage_check(300, 3, kernel="matern")
    stage_check(700, 6)
    if len(sys.argv) > 2:
        stage_check(int(sys.argv[1]), int(sys.argv[2]))
