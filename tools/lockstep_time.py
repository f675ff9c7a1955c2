"""Device time of B factorisations in lock step at N (HIP events inside the library): min / median of several samples."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, 8))
gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
ms = C.c_double()
for B in (1, 2, 4, 8):
    v = []
    for _ in range(7):
        if B == 1:
            gp._lib.bobe_debug_time_potrf(gp._h, 10, C.byref(ms))
        else:
            _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, B, 10, C.byref(ms)), "lockstep")
        v.append(ms.value)
    v = np.array(v)
    print(f"N={N} x{B}: min {v.min():.3f} ms  median {np.median(v):.3f} ms = {B*N**3/3/np.median(v)/1e9:.2f} TF/s ({B*N**3/3/np.median(v)/1e9/78.6*100:.1f} %)", flush=True)
