"""Where the host-side baseline spends its time (run on the GPU box's host cores)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.linalg import lapack
N, d = 4096, 8
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = rng.normal(size=N)
def t(f, reps=2):
    f(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps, r
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    Xs = torch.as_tensor(X / 0.6)
    def asm():
        sq = torch.zeros((N, N), dtype=torch.float64)
        for j in range(d):
            df = Xs[:, j][:, None] - Xs[:, j][None, :]
            sq.addcmul_(df, df)
        K = torch.exp(-0.5 * sq); K.diagonal().add_(1e-6); return K
    ta, K = t(asm)
    tc, L = t(lambda: torch.linalg.cholesky(K))
    ti, Ki = t(lambda: torch.cholesky_inverse(L))
    ts, al = t(lambda: torch.cholesky_solve(torch.as_tensor(y).reshape(-1, 1), L))
    def grad():
        WK = (al @ al.T - Ki) * K
        g = [float((WK * (Xs[:, j][:, None] - Xs[:, j][None, :]) ** 2).sum()) for j in range(d)]
        return g
    tg, _ = t(grad)
    print(f"torch threads {nt:3d}: assemble {ta:.3f}s  potrf {tc:.3f}s  potri {ti:.3f}s  potrs {ts:.3f}s  gradient {tg:.3f}s", flush=True)
Kn = K.numpy()
for nm, f in (("scipy dpotrf", lambda: lapack.dpotrf(Kn, lower=1, clean=1, overwrite_a=0)),):
    tt, (Ln, info) = t(f)
    print(f"{nm}: {tt:.3f}s")
tt, _ = t(lambda: lapack.dpotri(Ln, lower=1))
print(f"scipy dpotri: {tt:.3f}s")
try:
    from threadpoolctl import threadpool_info
    print([ (p.get('internal_api'), p.get('num_threads')) for p in threadpool_info()])
except Exception as e:
    print(e)
