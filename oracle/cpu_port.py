"""The benchmark cycle on the host cores — bench.py's ``cpu_baseline`` leg (kind "port").  TEST / MEASUREMENT
INFRASTRUCTURE: never imported by bobe_amd.

Same algorithm as ``bobe_oracle.cycle_value_and_grad`` / ``bobe_oracle.wip_sweep`` (the rank-1 sweep, not the
reference's O(C N^2 M) literal loop): LAPACK / BLAS through torch-CPU fp64 (MKL: dpotrf, dpotri, dpotrs, dtrsm, dgemm)
and the elementwise legs — kernel assembly and the d+1 gradient reductions — as fused OpenMP loops
(oracle/cpu_kernels.c).  Round 2 had those legs as torch expressions: 1.1 s of a 1.5 s evaluation at N = 4096 went into
their N x N temporaries while LAPACK took 0.4 s (profiles/r03_cpu_port_breakdown.txt); fused they take ~0.05 s.
tests/test_bench_cpu.py checks this module against the NumPy oracle.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import platform
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ck = None


def _kernels():
    """oracle/libbobe_cpu_kernels.so (``make -C oracle``; __graft_entry__.build() builds it)."""
    global _ck
    if _ck is None:
        path = os.path.join(_HERE, "libbobe_cpu_kernels.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "libbobe_cpu_kernels.so"], check=True)
        _ck = C.CDLL(path)
        _ck.ck_max_threads.restype = C.c_int
    return _ck


def _p(t):
    return C.c_void_p(t.data_ptr())


def _threads() -> int:
    import torch
    return int(torch.get_num_threads())


def rbf_sym(Xs, kvar, diag_add=0.0):
    """kvar exp(-r^2/2) on all pairs of the (scaled) rows of Xs, + diag_add on the diagonal: full symmetric matrix"""
    import torch
    n, d = Xs.shape
    K = torch.empty((n, n), dtype=torch.float64)
    _kernels().ck_rbf_sym(_p(Xs), C.c_int64(n), C.c_int(d), C.c_double(kvar), C.c_double(diag_add), _p(K), C.c_int(_threads()))
    return K


def rbf_rect(Xa, Xb, kvar):
    import torch
    out = torch.empty((Xa.shape[0], Xb.shape[0]), dtype=torch.float64)
    _kernels().ck_rbf_rect(_p(Xa), C.c_int64(Xa.shape[0]), _p(Xb), C.c_int64(Xb.shape[0]), C.c_int(Xa.shape[1]),
                           C.c_double(kvar), _p(out), C.c_int(_threads()))
    return out

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
            "threads": int(torch.get_num_threads()),
            "blas": f"torch-CPU fp64 ({blas} LAPACK/BLAS) + fused OpenMP elementwise legs (oracle/cpu_kernels.c)"}


def cycle_value_and_grad(X, y, ls, kvar, noise):
    """One value+grad of the RBF data-term MLL: fused assembly, dpotrf, dpotrs, dpotri, one fused pass for the d+1
    gradient reductions."""
    import torch
    X = torch.as_tensor(np.ascontiguousarray(X), dtype=torch.float64)
    yv = torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(-1, 1)
    n, d = X.shape
    Xs = (X / torch.as_tensor(np.asarray(ls, dtype=np.float64))).contiguous()
    K = rbf_sym(Xs, float(kvar), float(noise))
    L, info = torch.linalg.cholesky_ex(K)
    if int(info) != 0:
        return float("nan"), np.full(d + 1, np.nan)
    alpha = torch.cholesky_solve(yv, L)
    mll = float(-0.5 * (yv * alpha).sum() - torch.log(L.diagonal()).sum() - 0.5 * n * LOG_2PI)
    Kinv = torch.cholesky_inverse(L).contiguous()
    g = np.empty(d + 1)
    al = alpha.reshape(-1).contiguous()
    _kernels().ck_grad_rbf(_p(Xs), C.c_int64(n), C.c_int(d), _p(al), _p(Kinv), _p(K), C.c_double(noise),
                           g.ctypes.data_as(C.c_void_p), C.c_int(_threads()))
    return mll, g


def factor(X, y, ls, kvar, noise):
    """K, L, alpha at fixed hyper-parameters (recompute_cholesky, gp.py:544-550)."""
    import torch
    X = torch.as_tensor(np.ascontiguousarray(X), dtype=torch.float64)
    yv = torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(-1, 1)
    Xs = (X / torch.as_tensor(np.asarray(ls, dtype=np.float64))).contiguous()
    K = rbf_sym(Xs, float(kvar), float(noise))
    L = torch.linalg.cholesky(K)
    return {"Xs": Xs, "L": L, "alpha": torch.cholesky_solve(yv, L), "ls": np.asarray(ls, dtype=np.float64),
            "kvar": float(kvar), "noise": float(noise)}


def wip_sweep(f, cand, Z, y_std=1.0, chunk=4096):
    """Rank-1 WIPV / WIPStd sweep + posterior mean / variance of every candidate (bobe_oracle.wip_sweep)."""
    import torch
    ls = torch.as_tensor(f["ls"])
    L, kself = f["L"], f["kvar"] + f["noise"]
    Zs = (torch.as_tensor(np.ascontiguousarray(Z), dtype=torch.float64) / ls).contiguous()
    kz = rbf_rect(f["Xs"], Zs, f["kvar"])
    VZ = torch.linalg.solve_triangular(L, kz, upper=False)
    base = kself - (VZ * VZ).sum(0)
    C_ = cand.shape[0]
    out = {k: np.empty(C_) for k in ("mean", "var", "wipv", "wipstd")}
    for s in range(0, C_, chunk):
        Cs = (torch.as_tensor(np.ascontiguousarray(cand[s:s + chunk]), dtype=torch.float64) / ls).contiguous()
        kc = rbf_rect(f["Xs"], Cs, f["kvar"])
        vc = torch.linalg.solve_triangular(L, kc, upper=False)
        sc = kself - (vc * vc).sum(0)
        cross = rbf_rect(Cs, Zs, f["kvar"]) - vc.T @ VZ
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
