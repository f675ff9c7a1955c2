import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from bobe_amd import GP
rng = np.random.default_rng(0)
X = rng.uniform(size=(60, 2)); y = -np.sum((X - 0.5) ** 2, axis=1)
t0=time.perf_counter(); gp = GP(X, y, noise=1e-6); print("create+factor %.1f ms" % ((time.perf_counter()-t0)*1e3))
ls=np.array([0.5,0.5])
for i in range(3):
    t0=time.perf_counter(); gp.mll_data(ls,1.0,slot=0); print("slot0 eval %d: %.2f ms" % (i,(time.perf_counter()-t0)*1e3))
for s in (1,2,3):
    t0=time.perf_counter(); gp.mll_data(ls,1.0,slot=s); print("slot%d first eval: %.2f ms" % (s,(time.perf_counter()-t0)*1e3))
t0=time.perf_counter(); gp2 = GP(X, y, noise=1e-6); gp2.mll_data(ls,1.0,slot=0); print("second GP create + first slot eval %.1f ms" % ((time.perf_counter()-t0)*1e3))
