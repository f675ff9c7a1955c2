"""Parity in the numerical regime the reference actually runs in: a conditioning ladder.

Reference default ``noise = 1e-8`` (gp.py:201), BO-loop training sets of 600 ... 1800 clustered points, and the
hyper-parameters its fits reach on the 10-D Rosenbrock run of BASELINE config 5 (length scales 0.7 ... 3.8 in unit-cube
coordinates, kernel variance 10 ... 2.5e6: the values are the checkpoints of profiles/r04_config5.txt).  There cond(K)
runs from ~1e9 to beyond 1e16 and no fixed tolerance separates "wrong" from "as good as fp64 allows".  So three
implementations are compared on identical inputs:

  (i)   the HIP path - which forms v = L^-1 k with an EXPLICIT inverse factor (k_trtri_* -> k_trimul) where the reference
        does a triangular solve (gp.py:462, 571) and switches to a blocked forward substitution with 128-row diagonal
        blocks (k_blk_step) where the factor's (kvar + noise) / smallest pivot exceeds 1e6 (bobe_gp_set_refine_kappa) -
        the plain product, recorded beside it as 'raw', loses the fantasy variance from kernel variances of ~5e4 on
        (WIPV 1e-2 ... 1 off); larger diagonal blocks (256, 512) are recorded below the table;
  (ii)  the oracle's LAPACK / TRSM form (oracle/bobe_oracle.py: dpotrf + dtrsm, what jax.scipy lowers to on CPU);
  (iii) an extended-precision truth (oracle/bobe_oracle_xp.c: x87 long double, 64-bit significand, cross-checked against
        __float128 on the first rung)

and the assertion is   err(HIP vs truth) <= 4 x err(LAPACK vs truth) + floor   (floor: the fp64 parity tolerances of
SURVEY.md 8(d), ``TOL`` below) for the log marginal likelihood and its
gradient, posterior mean and variance, the fantasy variance var+(z|c), the WIPV / WIPStd scores; the chosen candidate
must be the truth's, or one the truth scores within that error of its best.  The table goes to
``gpurun_out/r06_conditioning.txt`` (committed as profiles/r06_conditioning.txt).

Deviation (vii) (the rank test) is switched OFF for the comparison (``pivot_floor_ulp = 0``: the reference's
sign-only rule) and its verdict at the default setting is recorded per rung.
"""
import os
import time

import numpy as np
import pytest
from scipy.linalg import solve_triangular
from scipy.stats import qmc

pytestmark = pytest.mark.gpu

D = 10
FLOOR = 1e-12          # gp.py:16


def _rosen10(x):
    return -np.sum(100.0 * (x[..., 1:] - x[..., :-1] ** 2) ** 2 + (1.0 - x[..., :-1]) ** 2, axis=-1) / 20.0


def _bo_like_design(n, seed=0):
    """A training set with the character of a BO run on config 5's likelihood (bounds [-2, 2]^10, unit-cube coordinates):
    64 scrambled-Sobol points, then states of tempered random-walk Metropolis chains on the likelihood (T = 1, 4, 16:
    a BO design crowds the posterior bulk and thins out around it).  Deterministic on the CPU; also returns posterior
    samples (T = 1) that are NOT in the design, as integration / query points."""
    rng = np.random.default_rng(seed)
    pts = [qmc.Sobol(D, scramble=True, seed=seed).random(64)]
    spare = []
    per = (n - 64 + 2) // 3
    for T in (1.0, 4.0, 16.0):
        u = np.full(D, 0.75)                                   # x = 1: the maximum
        lu = _rosen10(4 * u - 2)
        keep, step, it = [], 0.015 * np.sqrt(T), 0
        while len(keep) < per + (256 if T == 1.0 else 0):
            it += 1
            prop = u + step * rng.standard_normal(D)
            if np.all((prop > 0) & (prop < 1)):
                lp = _rosen10(4 * prop - 2)
                if np.log(rng.uniform()) < (lp - lu) / T:
                    u, lu = prop, lp
                    if it % 7 == 0:
                        keep.append(u.copy())
        pts.append(np.array(keep[:per]))
        if T == 1.0:
            spare = np.array(keep[per:])
    X = np.vstack(pts)[:n]
    return X, _rosen10(4 * X - 2), spare


# (N, kernel, length scales, kernel variance): the first nine dimensions share the first number, as the fits of the run did
LADDER = [
    (600, "rbf", [0.70, 0.79, 0.84, 0.87, 0.78, 0.80, 0.83, 0.74, 0.69, 3.10], 10.6),
    (1200, "rbf", [1.25] * 9 + [5.0], 432.0),
    (1800, "rbf", [2.31, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 5.0], 4.67e4),
    (1800, "rbf", [2.99, 2.94, 2.93, 2.93, 2.93, 2.93, 2.93, 2.93, 2.93, 5.0], 3.5e5),
    (1800, "rbf", [3.35, 3.28, 3.28, 3.28, 3.28, 3.29, 3.29, 3.29, 3.30, 5.0], 8.51e5),
    (1800, "rbf", [3.80, 3.73, 3.72, 3.73, 3.73, 3.73, 3.74, 3.74, 3.75, 5.0], 2.48e6),
    (600, "matern", [0.70, 0.79, 0.84, 0.87, 0.78, 0.80, 0.83, 0.74, 0.69, 3.10], 10.6),
    (1800, "matern", [2.31, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 2.27, 5.0], 4.67e4),
    (1800, "matern", [3.35, 3.28, 3.28, 3.28, 3.28, 3.29, 3.29, 3.29, 3.30, 5.0], 1.0e6),
]
NOISE = 1e-8
# the floor of the assertion: the fp64 parity tolerances of SURVEY.md 8(d) (|dLML| / |LML| <= 1e-10, gradient 1e-8, mean 1e-8,
# variance 1e-9 relative ... 1e-7) - an error inside them passes whatever LAPACK's happens to be on a well-conditioned rung
TOL = {"mll": 1e-10, "grad": 1e-8, "mean": 1e-8, "var": 1e-9, "fantasy": 1e-9, "wipv": 1e-7, "wipstd": 1e-7}
BLOCKS = [int(b) for b in os.environ.get("BOBE_LADDER_BLOCKS", "256,512").split(",") if b]
_ROWS = []


def _floored(v):
    v = np.where(np.isnan(v), FLOOR, v)
    return np.where(v < FLOOR, FLOOR, v)


def _quantities(mll, grad, mean, var, fant, y_std):
    """The compared quantities from (standardised, unfloored) ingredients, floors applied as the reference does."""
    f = _floored(fant) * y_std ** 2
    return {"mll": mll, "grad": grad, "mean": mean, "var": _floored(var), "fantasy": _floored(fant),
            "wipv": np.mean(f, axis=1), "wipstd": np.mean(np.sqrt(f), axis=1)}


def _err(a, truth, scale=None):
    a, truth = np.asarray(a, dtype=float), np.asarray(truth, dtype=float)
    s = np.max(np.abs(truth)) if scale is None else scale
    d = np.abs(a - truth)
    return float(np.max(np.where(np.isnan(d), np.inf, d)) / s)


@pytest.mark.parametrize("rung", range(len(LADDER)), ids=[f"N{n}_{k}_kvar{kv:g}" for n, k, _, kv in LADDER])
def test_conditioning_ladder(rung):
    from bobe_amd import GP
    from oracle import bobe_oracle as O
    from oracle import c_binding as CB
    n, kernel, ls, kvar = LADDER[rung]
    ls = np.array(ls)
    kid = 0 if kernel == "rbf" else 1
    X, y, spare = _bo_like_design(n)
    rng = np.random.default_rng(100 + rung)
    near = np.clip(X[rng.choice(n, 32, replace=False)] + 0.02 * rng.standard_normal((32, D)), 0.0, 1.0)
    cand = np.vstack([near, spare[:32], qmc.Sobol(D, scramble=True, seed=5).random(32)])
    Z = spare[64:128]
    C = cand.shape[0]

    # (iii) truth
    og = O.OracleGP(X, y, noise=NOISE, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    ys = np.asarray(og.train_y).reshape(-1)
    t0 = time.time()
    tr = CB.gp_truth(kid, X, ys, ls, kvar, NOISE, cand, Z)
    t_truth = time.time() - t0
    assert tr["info"] == 0 and tr["digits"] >= 64
    if rung == 0:                                               # the truth's own check: __float128 says the same
        tq = CB.gp_truth(kid, X, ys, ls, kvar, NOISE, cand, Z, kind="xq")
        assert tq["digits"] == 113 and abs(tq["mll"] - tr["mll"]) <= 1e-13 * abs(tq["mll"])
        assert _err(tr["grad"], tq["grad"]) < 1e-12 and _err(tr["fantasy"], tq["fantasy"], kvar) < 1e-16
    T = _quantities(tr["mll"], tr["grad"], tr["mean"], tr["var"], tr["fantasy"], og.y_std)
    kappa = (kvar * n) / tr["min_pivot"]                        # a cheap lower bound of cond(K): trace-scale / smallest pivot

    # (ii) LAPACK / TRSM form
    lapack_ok = bool(np.all(np.isfinite(og.cholesky)))
    if lapack_ok:
        mll_o, g_o = O.mll_value_and_grad(kernel, X, ys, ls, kvar, NOISE)
        L = og.cholesky
        kself = kvar + NOISE
        vc = solve_triangular(L, og._k12(cand), lower=True, check_finite=False)
        vz = solve_triangular(L, og._k12(Z), lower=True, check_finite=False)
        sc = kself - np.sum(vc * vc, axis=0)
        cross = og.kernel(cand, Z, ls, kvar, NOISE, include_noise=False) - vc.T @ vz
        with np.errstate(all="ignore"):
            fant_o = (kself - np.sum(vz * vz, axis=0))[None, :] - cross * cross / sc[:, None]
        fant_o = np.where(sc[:, None] >= 0, fant_o, np.nan)
        Oq = _quantities(mll_o, g_o, og._k12(cand).T @ og.alphas.reshape(-1), sc, fant_o, og.y_std)

    # (i) the HIP path, at its default setting and with the reference's sign-only rule
    gp = GP(X, y, noise=NOISE, kernel=kernel, lengthscales=ls, kernel_variance=kvar)
    rank_test_refuses = bool(gp.not_pd)
    gp.pivot_floor_ulp = 0.0
    gp.refine_kappa = -1.0                                     # first the plain product with the inverse factor, for the record
    gp.recompute_cholesky()
    raw = None
    if not gp.not_pd:
        assert not gp.refining
        sw_raw = gp.wip_sweep(cand, Z, want_mean_var=True)
        raw = {"wipv": sw_raw["wipv"], "wipstd": sw_raw["wipstd"], "fantasy": gp.fantasy_var(cand, Z) / og.y_std ** 2}
    gp.refine_kappa = 1e6                                      # the default
    gp.recompute_cholesky()
    hip_ok = not gp.not_pd
    row = {"rung": rung, "N": n, "kernel": kernel, "kvar": kvar, "ls0": float(ls[0]), "kappa": kappa,
           "min_pivot": tr["min_pivot"], "lapack_ok": lapack_ok, "hip_ok": hip_ok,
           "rank_test_refuses": rank_test_refuses, "t_truth": t_truth, "refined": bool(hip_ok and gp.refining)}
    assert hip_ok or not lapack_ok, "the HIP factorisation fails where LAPACK's passes"
    if hip_ok:
        th = np.log(np.append(ls, kvar))
        f, g = gp.neg_mll_value_and_grad(th)                  # f = -(mll + log prior); the default prior is flat: a constant
        lp_const = float(gp.prior_func(ls, kvar))
        sw = gp.wip_sweep(cand, Z, want_mean_var=True)
        fant_h = gp.fantasy_var(cand, Z) / og.y_std ** 2
        H = {"mll": -f - lp_const, "grad": -np.asarray(g)[:D + 1], "mean": sw["mean"], "var": sw["var"],
             "fantasy": fant_h, "wipv": sw["wipv"], "wipstd": sw["wipstd"]}
        # the score VALUES of the gradient entry point (acquisition.py:403-412 refines with them): the few-candidate
        # matrix-vector path (<= 16 candidates) and the batched one must be as close to the truth as the sweep's
        for label, idx in (("few", np.arange(3)), ("batched", np.arange(20))):
            wv, ws, _, _ = gp.wip_grad(cand[idx], Z)
            row["hip_wipv_grad_" + label] = _err(wv, T["wipv"][idx], np.max(np.abs(T["wipv"])))
            row["hip_wipstd_grad_" + label] = _err(ws, T["wipstd"][idx], np.max(np.abs(T["wipstd"])))
    scales = {"mll": None, "grad": None, "mean": None, "var": kvar + NOISE, "fantasy": kvar + NOISE, "wipv": None,
              "wipstd": None}
    # the same quantities through the blocked forward substitution (bobe_gp_set_solve_block) at several block heights
    if hip_ok:
        keep = gp.solve_block
        for b in BLOCKS:
            gp.solve_block = b
            gp.recompute_cholesky()
            swb = gp.wip_sweep(cand, Z, want_mean_var=True)
            fb = gp.fantasy_var(cand, Z) / og.y_std ** 2
            for q, val in (("var", swb["var"]), ("fantasy", fb), ("wipv", swb["wipv"]), ("wipstd", swb["wipstd"])):
                row[f"blk{b}_{q}"] = _err(val, T[q], scales[q])
            row[f"blk{b}_argmin"] = (int(swb["argmin_v"]) == int(np.argmin(T["wipv"])),
                                     int(swb["argmin_s"]) == int(np.argmin(T["wipstd"])))
        gp.solve_block = keep
        gp.recompute_cholesky()
    for q in ("fantasy", "wipv", "wipstd"):
        row["raw_" + q] = _err(raw[q], T[q], scales[q]) if raw is not None else float("nan")
    for q, sc_ in scales.items():
        row["hip_" + q] = _err(H[q], T[q], sc_) if hip_ok else float("nan")
        row["lap_" + q] = _err(Oq[q], T[q], sc_) if lapack_ok else float("nan")
    # the chosen candidate (acquisition.py:397): the truth's, or one the truth scores within the fp64 error of its best
    for key, name in (("wipv", "argmin_v"), ("wipstd", "argmin_s")):
        it = int(np.argmin(T[key]))
        if hip_ok:
            ih = int(sw[name])
            row["hip_" + name] = ih == it
            row["gap_" + name] = float((T[key][ih] - T[key][it]) / T[key][it])
        if lapack_ok:
            row["lap_" + name] = int(np.argmin(Oq[key])) == it
    _ROWS.append(row)
    _write_table()
    if not (hip_ok and lapack_ok):
        return
    for q in scales:
        assert row["hip_" + q] <= 4.0 * row["lap_" + q] + TOL[q], (q, row["hip_" + q], row["lap_" + q])
    for key, name in (("wipv", "argmin_v"), ("wipstd", "argmin_s")):
        assert row["hip_" + name] or row["gap_" + name] <= 4.0 * row["lap_" + key] + TOL[key], (name, row)
    for label in ("few", "batched"):
        for key in ("wipv", "wipstd"):
            assert row[f"hip_{key}_grad_{label}"] <= 4.0 * row["lap_" + key] + TOL[key], (label, key, row)


def _write_table():
    out = os.environ.get("BOBE_CONDITIONING_OUT", os.path.join("gpurun_out", "r06_conditioning.txt"))
    try:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    except OSError:
        return
    qs = ["mll", "grad", "mean", "var", "fantasy", "wipv", "wipstd"]
    lines = ["# conditioning ladder (tests/test_gpu_conditioning.py): noise 1e-8, BO-like design on the 10-D Rosenbrock "
             "likelihood, hyper-parameters of the config-5 checkpoints",
             "# errors against the extended-precision truth (x87 long double); 'hip' = HIP path (explicit inverse factor, rank "
             "test off), 'lap' = LAPACK dpotrf + dtrsm form",
             "# mll / grad / mean / wipv / wipstd: max |delta| / max |truth|;  var / fantasy: max |delta| / (kvar + noise)",
             "# kappa = N kvar / smallest pivot (a lower bound of cond K);  rank64 = the default rank test (64 ulp) refuses the "
             "rung;  argmin columns: picks the truth's candidate (gap = relative score excess of the pick)",
             "# refine = the HIP path solved for v by blocked forward substitution (default threshold 1e6, 128-row diagonal blocks); raw_* = the same "
             "quantity WITHOUT it (bobe_gp_set_refine_kappa(-1))", ""]
    hdr = f"{'rung':<28}{'kappa':>9}{'minpiv':>9} {'lapack':>6} {'rank64':>6} {'refine':>6} " + " ".join(f"{'hip_' + q:>11}{'lap_' + q:>11}" for q in qs) \
        + f" {'raw_fantasy':>11} {'raw_wipv':>9} {'raw_wipstd':>10} {'argv h/l':>9} {'args h/l':>9} {'gap_v':>8} {'gap_s':>8}"
    lines.append(hdr)
    for r in _ROWS:
        name = f"N{r['N']}_{r['kernel']}_ls{r['ls0']:g}_kv{r['kvar']:g}"
        cells = " ".join(f"{r['hip_' + q]:>11.2e}{r['lap_' + q]:>11.2e}" for q in qs)
        lines.append(f"{name:<28}{r['kappa']:>9.1e}{r['min_pivot']:>9.1e} {str(r['lapack_ok']):>6} {str(r['rank_test_refuses']):>6} "
                     f"{str(r['refined']):>6} {cells} {r['raw_fantasy']:>11.2e} {r['raw_wipv']:>9.2e} {r['raw_wipstd']:>10.2e} {str(r.get('hip_argmin_v', '-'))[0]}/{str(r.get('lap_argmin_v', '-'))[0]:<7} "
                     f"{str(r.get('hip_argmin_s', '-'))[0]}/{str(r.get('lap_argmin_s', '-'))[0]:<7} "
                     f"{r.get('gap_argmin_v', float('nan')):>8.1e} {r.get('gap_argmin_s', float('nan')):>8.1e}")
    lines += ["", "# larger diagonal blocks of the substitution (bobe_gp_set_solve_block = b) on the same rungs, beside the shipped default "
              "('hip': b = 128) and LAPACK's triangular solve ('lap'); argmin = (WIPV, WIPStd) picks the truth's candidate"]
    qb = ["var", "fantasy", "wipv", "wipstd"]
    for r in _ROWS:
        name = f"N{r['N']}_{r['kernel']}_ls{r['ls0']:g}_kv{r['kvar']:g}"
        lines.append(f"{name:<28} " + " ".join(f"{'hip_' + q:>8}={r['hip_' + q]:<9.2e}{'lap':>4}={r['lap_' + q]:<9.2e}" for q in qb))
        for b in BLOCKS:
            if f"blk{b}_var" in r:
                lines.append(f"{'  b=' + str(b):<28} " + " ".join(f"{'blk_' + q:>8}={r[f'blk{b}_{q}']:<9.2e}{'':>14}" for q in qb)
                             + f" argmin {r[f'blk{b}_argmin']}")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
