"""Timing probe for experiments that are switched by environment variables (read once per process): device time of the
lock-step factorisation (B = 1, 4) and wall time of a lock-step batch of four value+gradient evaluations at N = 4096.
  BOBE_FILLER_ITERS=50 python tools/filler_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402

N, d = int(os.environ.get("PROBE_N", "4096")), 8
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(d, 0.6))
lib, h = gp._lib, gp._h
ms = C.c_double()
out = []
for B in ((1, 4) if not os.environ.get("PROBE_B1") else (1,)):
    _lib.check(lib.bobe_debug_time_potrf_lockstep(h, B, 10, C.byref(ms)), "potrf_lockstep")
    out.append(f"potrf lock-step B={B}: {ms.value:.3f} ms")
ls = np.full((4, d), 0.55) + 0.01 * np.arange(4)[:, None]
m, g = gp.mll_data_batch(ls, np.ones(4))
if not os.environ.get("PROBE_B1"):
    t0 = time.perf_counter()
    for _ in range(10):
        m, g = gp.mll_data_batch(ls, np.ones(4))
    out.append(f"eval batch B=4: {(time.perf_counter() - t0) * 100:.3f} ms   mll[0]={m[0]:.6f}")
t0 = time.perf_counter()
for _ in range(10):
    m1, g1 = gp.mll_data(ls[0], 1.0)
out.append(f"eval alone: {(time.perf_counter() - t0) * 100:.3f} ms   same bits as batch: {m1 == m[0] and np.array_equal(g1, g[0])}")
t0 = time.perf_counter()
for _ in range(10):
    gp.recompute_cholesky()
out.append(f"factor: {(time.perf_counter() - t0) * 100:.3f} ms")
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("BOBE_"))
print(f"[{tag or 'default'}] " + " | ".join(out), flush=True)
