"""Per-kernel-class device time of one value+gradient evaluation (HIP events inside the library) on the GPU box.
usage: python tools/chol_classes.py [N]   (tuning through the BOBE_* environment variables)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402
from bobe_amd.gp import GP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 8
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
y = np.sin(X.sum(1))
gp = GP(X, y, noise=1e-4, lengthscales=np.full(d, 0.6))
ls = np.full(d, 0.55)
gp.mll_data(ls, 1.0)
out = []
for name, tag in _lib.PROF.items():
    if name in ("trimul", "cross"):
        continue
    gp._lib.bobe_gp_profile_select(gp._h, tag)
    reps = 3
    for _ in range(reps):
        gp.mll_data(ls, 1.0)
    ms, n = C.c_double(), C.c_int64()
    gp._lib.bobe_gp_profile_read(gp._h, C.byref(ms), C.byref(n))
    out.append(f"{name} {ms.value / reps:.3f} ms/{n.value // reps}")
gp._lib.bobe_gp_profile_select(gp._h, 0)
pm = C.c_double()
gp._lib.bobe_debug_time_potrf(gp._h, 3, C.byref(pm))
print(f"N={N} " + " | ".join(out) + f" | potrf {pm.value:.3f} ms", flush=True)
