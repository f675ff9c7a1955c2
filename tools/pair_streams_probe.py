"""Probe: the fit's four restarts as two lock-step pairs on two handles (two streams) vs four in lock step / four slots."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP
N = 4096
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, 8)); y = np.sin(X.sum(1))
gps = [GP(X, y, noise=1e-4, lengthscales=np.full(8, 0.6)) for _ in range(2)]
ls = np.full((4, 8), 0.55) + 0.01 * np.arange(4)[:, None]
kv = np.ones(4)
def rounds_pairs(n):
    def work(i):
        for _ in range(n):
            gps[i].mll_data_batch(ls[2 * i:2 * i + 2], kv[:2])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
def rounds_lock(n):
    for _ in range(n):
        gps[0].mll_data_batch(ls, kv)
def rounds_slots(n):
    def work(s):
        for _ in range(n):
            gps[0].mll_data(ls[s], 1.0, slot=s)
    th = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
for name, f in (("2 pairs on 2 streams", rounds_pairs), ("4 in lock step", rounds_lock), ("4 slots", rounds_slots)):
    f(2)
    t0 = time.perf_counter(); f(5); dt = (time.perf_counter() - t0) / 5
    print(f"{name:24s}: {dt*1e3:.3f} ms per round of 4 evaluations", flush=True)
