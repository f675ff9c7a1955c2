import numpy as np, sys
sys.path.insert(0, "/root/repo")
from bobe_amd import GP
from bobe_amd.bo import gp_fit
rng = np.random.RandomState(42)
X = rng.uniform(size=(40, 2))
y = -np.sum((X - 0.5) ** 2, axis=1)
for conc in (False, True, True, True, True):
    gp = GP(X, y, noise=1e-6)
    gp.concurrent_restarts = conc
    r = gp_fit(gp, maxiters=50, n_restarts=3, rng=np.random.default_rng(5))
    print(conc, repr(r["mll"]), r["params"], flush=True)
