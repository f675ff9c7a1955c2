"""Factorisation rate at larger N: alone and in lock step (device time, HIP events)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib
from bobe_amd.gp import GP
for N in ([int(a) for a in sys.argv[1:]] or (6144, 8192, 12288, 16384, 24576, 32768)):
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, 8))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
    ms = C.c_double()
    gp._lib.bobe_debug_time_potrf(gp._h, 3, C.byref(ms))
    out = f"N={N:6d} alone {ms.value:8.3f} ms = {N**3/3/ms.value/1e9:6.2f} TF/s ({N**3/3/ms.value/1e9/78.6*100:4.1f} %)"
    for B in ((2, 4) if N <= 16384 else (2,)):
        _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, B, 2, C.byref(ms)), "lockstep")
        out += f" | x{B} {ms.value:8.3f} ms = {B*N**3/3/ms.value/1e9:6.2f} TF/s ({B*N**3/3/ms.value/1e9/78.6*100:4.1f} %)"
    import time
    t0 = time.perf_counter()
    m, g = gp.mll_data(np.full(8, 0.55), 1.0)
    dt = time.perf_counter() - t0
    out += f" | value+grad {dt*1e3:8.1f} ms = {N**3/dt/1e12:5.1f} TF/s"
    print(out, flush=True)
    del gp
