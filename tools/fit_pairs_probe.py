"""Fit of 4 restarts x 5 evaluations at the headline size: one lock-step batch of four per round (GP.fit's way) against TWO
lock-step batches of two on two handles / streams, each pair advancing without waiting for the other."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP
from bobe_amd.synthetic import synthetic_problem, theta_schedule

N, d = 4096, 8
X, y, _, _ = synthetic_problem(N, d, 8, 8, noise=1e-6)
th = np.array(theta_schedule(d))                                  # 20 log-hyper-parameter vectors
ls_all, kv_all = np.exp(th[:, :d]), np.exp(th[:, d])
gp = GP(X, y, noise=1e-6, lengthscales=np.full(d, 0.6))
gp2 = gp.copy()


def rounds(g, idx_rounds, out):
    for idx in idx_rounds:
        m, gr = g.mll_data_batch(ls_all[idx], kv_all[idx])
        out[tuple(idx)] = (m, gr)


def four():
    out = {}
    rounds(gp, [np.arange(4) + 4 * j for j in range(5)], out)
    return out


def two_two():
    out = {}
    ta = threading.Thread(target=rounds, args=(gp, [np.array([0, 1]) + 4 * j for j in range(5)], out))
    tb = threading.Thread(target=rounds, args=(gp2, [np.array([2, 3]) + 4 * j for j in range(5)], out))
    ta.start(); tb.start(); ta.join(); tb.join()
    return out


for name, fn in (("one batch of four", four), ("two batches of two", two_two), ("one batch of four", four), ("two batches of two", two_two)):
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: median {np.median(ts):.2f} ms (min {min(ts):.2f})", flush=True)
a, b = four(), two_two()
m4 = np.concatenate([a[tuple(np.arange(4) + 4 * j)][0] for j in range(5)])
m2 = np.concatenate([np.concatenate([b[(4 * j, 4 * j + 1)][0], b[(4 * j + 2, 4 * j + 3)][0]]) for j in range(5)])
print("same bits:", np.array_equal(m4, m2))
