"""Lock-step / lone factorisation time at N (argv[1]) under the current BOBE_* tuning environment."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib
from bobe_amd.gp import GP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, 8))
gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
ms = C.c_double()
gp._lib.bobe_debug_time_potrf(gp._h, 5, C.byref(ms))
out = f"x1 {ms.value:.3f}"
for B in (2, 4, 8):
    _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, B, 4, C.byref(ms)), "lockstep")
    out += f" | x{B} {ms.value:.3f} ms ({B*N**3/3/ms.value/1e9/78.6*100:.1f} %)"
print({k: v for k, v in os.environ.items() if k.startswith("BOBE_")}, out, flush=True)
