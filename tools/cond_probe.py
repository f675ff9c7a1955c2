import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ctypes as C
import numpy as np
from scipy.linalg import solve_triangular
import test_gpu_conditioning as T
from bobe_amd import GP, _lib
from oracle import bobe_oracle as O, c_binding as CB
from scipy.stats import qmc
rung = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, kernel, ls, kvar = T.LADDER[rung]; ls = np.array(ls)
X, y, spare = T._bo_like_design(n)
rng = np.random.default_rng(100 + rung)
near = np.clip(X[rng.choice(n, 32, replace=False)] + 0.02 * rng.standard_normal((32, 10)), 0, 1)
cand = np.vstack([near, spare[:32], qmc.Sobol(10, scramble=True, seed=5).random(32)]); Z = spare[64:128]
og = O.OracleGP(X, y, noise=1e-8, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
ys = np.asarray(og.train_y).reshape(-1)
tr = CB.gp_truth(0, X, ys, ls, kvar, 1e-8, cand, Z)
gp = GP(X, y, noise=1e-8, kernel=kernel, lengthscales=ls, kernel_variance=kvar); gp.pivot_floor_ulp = 0; gp.recompute_cholesky()
Lh = gp.cholesky
Linv = np.empty((n, n)); _lib.check(gp._lib.bobe_debug_linv(gp._h, _lib.ptr(Linv)), "linv")
kc, kz = og._k12(cand), og._k12(Z)
kcz = og.kernel(cand, Z, ls, kvar, 1e-8, include_noise=False)
kself = kvar + 1e-8
def pieces(vc, vz):
    sc = kself - np.sum(vc * vc, 0); bz = kself - np.sum(vz * vz, 0); cr = kcz - vc.T @ vz
    return sc, bz, cr, bz[None, :] - cr * cr / sc[:, None]
def report(name, vc, vz):
    sc, bz, cr, f = pieces(vc, vz)
    F = lambda v: np.where(np.isnan(v) | (v < 1e-12), 1e-12, v)
    wv = np.mean(F(f), 1); wt = np.mean(F(tr["fantasy"]), 1)
    print(f"{name:28s} s_c {np.max(np.abs(sc-tr['var']))/kvar:.2e} base_z {np.max(np.abs(bz-tr['var_z']))/kvar:.2e} cross {np.max(np.abs(cr-tr['cross']))/kvar:.2e} "
          f"fantasy {np.max(np.abs(F(f)-F(tr['fantasy'])))/kvar:.2e} wipv {np.max(np.abs(wv-wt))/np.max(wt):.2e}  resid |L v - k|/|k| {np.max(np.abs(Lh@vc-kc))/np.max(np.abs(kc)):.1e}")
Lo = og.cholesky
report("LAPACK L, TRSM", solve_triangular(Lo, kc, lower=True), solve_triangular(Lo, kz, lower=True))
report("HIP L, TRSM", solve_triangular(Lh, kc, lower=True), solve_triangular(Lh, kz, lower=True))
report("HIP Linv @ k (numpy)", Linv @ kc, Linv @ kz)
# one step of iterative refinement of v with the factor: v += Linv (k - L v)
def refine(v, k): return v + Linv @ (k - Lh @ v)
report("HIP Linv @ k + 1 refinement", refine(Linv @ kc, kc), refine(Linv @ kz, kz))
sw = gp.wip_sweep(cand, Z, want_mean_var=True); fh = gp.fantasy_var(cand, Z) / og.y_std ** 2
F = lambda v: np.where(np.isnan(v) | (v < 1e-12), 1e-12, v)
print("HIP sweep: s_c %.2e fantasy %.2e wipv %.2e" % (np.max(np.abs(F(sw["var"]) - F(tr["var"]))) / kvar, np.max(np.abs(fh - F(tr["fantasy"]))) / kvar,
      np.max(np.abs(sw["wipv"] / og.y_std**2 - np.mean(F(tr["fantasy"]), 1))) / np.max(np.mean(F(tr["fantasy"]), 1))))
print("truth fantasy range", tr["fantasy"].min(), np.median(tr["fantasy"]), tr["fantasy"].max(), "var_z", tr["var_z"].min(), tr["var_z"].max(), "cond(Linv) max", np.abs(Linv).max())
