"""Lock-step value+gradient batch time at N=4096 under the current tuning environment variables."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, 8))
gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
out = []
for B in (4, 8):
    ls = np.full((B, 8), 0.55) + 0.01 * np.arange(B)[:, None]
    gp.mll_data_batch(ls, np.ones(B))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); gp.mll_data_batch(ls, np.ones(B)); ts.append(time.perf_counter() - t0)
    out.append(f"B={B}: {min(ts)*1e3:.3f} ms ({min(ts)*1e3/B:.3f}/eval)")
print({k: v for k, v in os.environ.items() if k.startswith("BOBE_")}, " | ".join(out), flush=True)
