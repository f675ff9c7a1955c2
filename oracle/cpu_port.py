"""The benchmark cycle on the host cores — bench.py's ``cpu_baseline`` leg (kind "port").  TEST / MEASUREMENT
INFRASTRUCTURE: never imported by bobe_amd.

Same algorithm as ``bobe_oracle.cycle_value_and_grad`` / ``bobe_oracle.wip_sweep`` (the rank-1 sweep, not the
reference's O(C N^2 M) literal loop), restated on torch-CPU fp64 so that the elementwise N^2 / N*C passes run on all
host threads like the LAPACK / BLAS calls do (the NumPy form spends most of its time in single-threaded ufuncs).
tests/test_oracle.py checks it against the NumPy oracle.
"""
from __future__ import annotations

import math
import platform

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)
FLOOR = 1e-12


def host_description() -> dict:
    import torch
    model = platform.processor() or platform.machine()
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas = "unknown"
    try:
        cfg = torch.__config__.show()
        blas = "MKL" if "USE_MKL=ON" in cfg or "BLAS_INFO=mkl" in cfg else ("OpenBLAS" if "open" in cfg.lower() else "generic")
    except Exception:
        pass
    return {"cpu_model": model, "logical_cpus": int(__import__("os").cpu_count() or 1),
            "threads": int(torch.get_num_threads()), "blas": f"torch-CPU fp64 ({blas} LAPACK/BLAS)"}


def _scaled_sqdist(Xs_a, Xs_b):
    """direct differences (dist_sq, gp.py:80-96), dimension by dimension: no (n1, n2, d) temporary"""
    import torch
    sq = torch.zeros((Xs_a.shape[0], Xs_b.shape[0]), dtype=torch.float64)
    for j in range(Xs_a.shape[1]):
        df = Xs_a[:, j][:, None] - Xs_b[:, j][None, :]
        sq.addcmul_(df, df)
    return sq


def cycle_value_and_grad(X, y, ls, kvar, noise):
    """One value+grad of the RBF data-term MLL: assembly, dpotrf, dpotrs, dpotri, d+1 fused N^2 reductions."""
    import torch
    X = torch.as_tensor(np.ascontiguousarray(X), dtype=torch.float64)
    yv = torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(-1, 1)
    n, d = X.shape
    Xs = X / torch.as_tensor(np.asarray(ls, dtype=np.float64))
    Kt = kvar * torch.exp(-0.5 * _scaled_sqdist(Xs, Xs))
    K = Kt.clone()
    K.diagonal().add_(noise)
    L, info = torch.linalg.cholesky_ex(K)
    if int(info) != 0:
        return float("nan"), np.full(d + 1, np.nan)
    alpha = torch.cholesky_solve(yv, L)
    mll = float(-0.5 * (yv * alpha).sum() - torch.log(L.diagonal()).sum() - 0.5 * n * LOG_2PI)
    Kinv = torch.cholesky_inverse(L)
    WK = (alpha @ alpha.T - Kinv) * Kt
    g = np.empty(d + 1)
    for j in range(d):
        df = Xs[:, j][:, None] - Xs[:, j][None, :]
        g[j] = 0.5 * float((WK * df * df).sum())
    g[d] = 0.5 * float(WK.sum())
    return mll, g


def factor(X, y, ls, kvar, noise):
    """K, L, alpha at fixed hyper-parameters (recompute_cholesky, gp.py:544-550)."""
    import torch
    X = torch.as_tensor(np.ascontiguousarray(X), dtype=torch.float64)
    yv = torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(-1, 1)
    Xs = X / torch.as_tensor(np.asarray(ls, dtype=np.float64))
    K = kvar * torch.exp(-0.5 * _scaled_sqdist(Xs, Xs))
    K.diagonal().add_(noise)
    L = torch.linalg.cholesky(K)
    return {"Xs": Xs, "L": L, "alpha": torch.cholesky_solve(yv, L), "ls": np.asarray(ls, dtype=np.float64),
            "kvar": float(kvar), "noise": float(noise)}


def wip_sweep(f, cand, Z, y_std=1.0, chunk=4096):
    """Rank-1 WIPV / WIPStd sweep + posterior mean / variance of every candidate (bobe_oracle.wip_sweep)."""
    import torch
    ls = torch.as_tensor(f["ls"])
    L, kself = f["L"], f["kvar"] + f["noise"]
    Zs = torch.as_tensor(np.ascontiguousarray(Z), dtype=torch.float64) / ls
    kz = f["kvar"] * torch.exp(-0.5 * _scaled_sqdist(f["Xs"], Zs))
    VZ = torch.linalg.solve_triangular(L, kz, upper=False)
    base = kself - (VZ * VZ).sum(0)
    C = cand.shape[0]
    out = {k: np.empty(C) for k in ("mean", "var", "wipv", "wipstd")}
    for s in range(0, C, chunk):
        Cs = torch.as_tensor(np.ascontiguousarray(cand[s:s + chunk]), dtype=torch.float64) / ls
        kc = f["kvar"] * torch.exp(-0.5 * _scaled_sqdist(f["Xs"], Cs))
        vc = torch.linalg.solve_triangular(L, kc, upper=False)
        sc = kself - (vc * vc).sum(0)
        cross = f["kvar"] * torch.exp(-0.5 * _scaled_sqdist(Cs, Zs)) - vc.T @ VZ
        var = base[None, :] - cross * cross / sc[:, None]
        var = torch.where(sc[:, None] >= 0, var, torch.full_like(var, float("nan")))
        var = torch.where(torch.isnan(var), torch.full_like(var, FLOOR), var).clamp_min(FLOOR) * (y_std ** 2)
        out["wipv"][s:s + chunk] = var.mean(1).numpy()
        out["wipstd"][s:s + chunk] = var.sqrt().mean(1).numpy()
        out["mean"][s:s + chunk] = (kc.T @ f["alpha"]).reshape(-1).numpy()
        pv = torch.where(torch.isnan(sc), torch.full_like(sc, FLOOR), sc).clamp_min(FLOOR)
        out["var"][s:s + chunk] = pv.numpy()
    out["argmin_v"] = int(np.argmin(out["wipv"]))
    out["argmin_s"] = int(np.argmin(out["wipstd"]))
    return out
