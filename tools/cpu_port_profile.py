"""Where the host-side baseline (oracle/cpu_port.py) spends its time, leg by leg (run on the GPU box's host cores)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from oracle import cpu_port as P  # noqa: E402

N, d = 4096, 8
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d))
y = rng.normal(size=N)


def t(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps, r


for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    Xs = torch.as_tensor(X / 0.6).contiguous()
    ta, K = t(lambda: P.rbf_sym(Xs, 1.0, 1e-6))
    tc, L = t(lambda: torch.linalg.cholesky(K))
    ti, Ki = t(lambda: torch.cholesky_inverse(L))
    ts, al = t(lambda: torch.cholesky_solve(torch.as_tensor(y).reshape(-1, 1), L))
    g = np.empty(d + 1)
    a1 = al.reshape(-1).contiguous()
    tg, _ = t(lambda: P._kernels().ck_grad_rbf(P._p(Xs), C.c_int64(N), C.c_int(d), P._p(a1), P._p(Ki), P._p(K), C.c_double(1e-6),
                                               g.ctypes.data_as(C.c_void_p), C.c_int(nt)))
    tv, _ = t(lambda: P.cycle_value_and_grad(X, y, np.full(d, 0.6), 1.0, 1e-6))
    print(f"threads {nt:3d}: assemble {ta:.3f}s  potrf {tc:.3f}s  potri {ti:.3f}s  potrs {ts:.3f}s  gradient {tg:.3f}s"
          f"  | value+grad {tv:.3f}s", flush=True)
