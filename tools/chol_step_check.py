"""Correctness + timing of the batched (lock-step) Cholesky on the GPU box.

  * L from bobe_gp_factor against LAPACK on the same K (several N incl. ragged and single-block sizes)
  * bobe_gp_mll_batch (lock-step pipeline) against one-at-a-time bobe_gp_mll: bitwise
  * device time of potrf alone, B in lock step, and B on private streams (the round-1 slots)
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402
from oracle import bobe_oracle as O  # noqa: E402  (checker only)

d = 8
ok = True
for N in (1, 17, 128, 129, 300, 641, 1024, 1500, 2048):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
    ls = np.full(d, 0.6)
    gp = GP(X, y, noise=1e-4, lengthscales=ls)
    K = O.rbf_kernel(X, X, ls, 1.0, 1e-4, True)
    Lr = np.linalg.cholesky(K)
    err = np.max(np.abs(gp.cholesky - Lr)) / np.max(np.abs(Lr))
    flag = "ok" if err < 1e-11 else "FAIL"
    ok &= err < 1e-11
    line = f"N={N:5d}  max|L-L_lapack|/max|L| = {err:.2e} {flag}"
    if N >= 2:
        B = 4
        lsb = np.full((B, d), 0.55) + 0.02 * np.arange(B)[:, None]
        kv = 1.0 + 0.1 * np.arange(B)
        mb, gb = gp.mll_data_batch(lsb, kv)
        same = True
        for b in range(B):
            m1, g1 = gp.mll_data(lsb[b], kv[b])
            same &= (m1 == mb[b]) and np.array_equal(g1, gb[b])
        ok &= bool(same)
        line += f"   batch==single bitwise: {same}"
    print(line, flush=True)
    del gp

for N in (1024, 2048, 4096, 8192):
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(X.sum(1))
    gp = GP(X, y, noise=1e-4, lengthscales=np.full(d, 0.6))
    ms = C.c_double()
    gp._lib.bobe_debug_time_potrf(gp._h, 5, C.byref(ms))
    out = f"N={N:5d} potrf x1 {ms.value:7.3f} ms = {N**3/3/ms.value/1e9:6.2f} TF/s |"
    for B in (2, 4, 8):
        _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, B, 3, C.byref(ms)), "lockstep")
        out += f" lockstep x{B} {ms.value:7.3f} ms = {B*N**3/3/ms.value/1e9:6.2f} TF/s |"
    if N <= 4096:
        for B in (4,):
            _lib.check(gp._lib.bobe_debug_time_potrf_batch(gp._h, B, 3, C.byref(ms)), "slots")
            out += f" streams x{B} {ms.value:7.3f} ms = {B*N**3/3/ms.value/1e9:6.2f} TF/s |"
    print(out, flush=True)
    lsb = np.full((8, d), 0.55) + 0.01 * np.arange(8)[:, None]
    for B in (1, 2, 4, 8):
        gp.mll_data_batch(lsb[:B], np.ones(B))
        t0 = time.perf_counter()
        for _ in range(3):
            gp.mll_data_batch(lsb[:B], np.ones(B))
        dt = (time.perf_counter() - t0) / 3
        print(f"        value+grad B={B}: {dt*1e3:8.3f} ms per batch, {dt*1e3/B:7.3f} ms per evaluation = "
              f"{B*N**3/dt/1e12:6.2f} TF/s", flush=True)
    del gp
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
