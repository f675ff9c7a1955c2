"""Cholesky / value+grad timing vs N on the GPU box (device time via HIP events inside the library)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402

for N in (512, 1024, 2048, 4096, 8192, 12288):
    d = 8
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(X.sum(1))
    gp = GP(X, y, noise=1e-4, lengthscales=np.full(d, 0.6))
    ms = C.c_double()
    gp._lib.bobe_debug_time_potrf(gp._h, 3, C.byref(ms))
    ls = np.full(d, 0.55)
    gp.mll_data(ls, 1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        gp.mll_data(ls, 1.0)
    vg = (time.perf_counter() - t0) / 3
    print(f"N={N:6d}  potrf {ms.value:8.3f} ms = {N**3/3/ms.value/1e9:7.2f} TFLOP/s   value+grad {vg*1e3:8.3f} ms = {N**3/vg/1e12:6.2f} TFLOP/s", flush=True)
    del gp
