"""Per-step timeline of the batched Cholesky from a rocprofv3 kernel trace.
  run   : rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_X -- python tools/chol_trace.py run N B
  parse : python tools/chol_trace.py parse <kernel_trace.csv> N B
"""
import csv
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(N, B):
    from bobe_amd import _lib
    from bobe_amd.gp import GP
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, 8))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
    ms = C.c_double()
    if B == 1:
        _lib.check(gp._lib.bobe_debug_time_potrf(gp._h, 3, C.byref(ms)), "potrf")
    else:
        _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, B, 3, C.byref(ms)), "lockstep")
    print(f"N={N} B={B}: {ms.value:.3f} ms = {B * N**3 / 3 / ms.value / 1e9:.2f} TF/s")


def parse(path, N, B):
    rows = list(csv.DictReader(open(path)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    ev.sort()
    nb = (N + 127) // 128
    steps = [e for e in ev if "k_chol_step" in e[2]]
    cols = [e for e in ev if "k_syrk_trail" in e[2]]
    # the last factorisation of the run: nb step launches, nb-1 column launches
    steps, cols = steps[-nb:], cols[-(nb - 1):]
    t0 = steps[0][0]
    print(f"{'k':>3} {'gap_us':>7} {'col_us':>7} {'gap_us':>7} {'step_us':>8} {'upd GF':>8} {'upd TF/s if bound':>18}")
    tot_step = tot_col = tot_gap = 0.0
    prev_end = None
    for k in range(nb):
        st = steps[k]
        rem = nb - 1 - k
        flops = B * 128.0 * (rem * 128.0) ** 2 if k > 0 else 0.0      # lower tiles right of column k, K = 128
        if k > 0:
            c = cols[k - 1]
            g1 = (c[0] - prev_end) / 1e3
            cu = (c[1] - c[0]) / 1e3
            g2 = (st[0] - c[1]) / 1e3
        else:
            g1 = cu = g2 = 0.0
        su = (st[1] - st[0]) / 1e3
        tot_step += su
        tot_col += cu
        tot_gap += g1 + g2
        prev_end = st[1]
        print(f"{k:3d} {g1:7.2f} {cu:7.2f} {g2:7.2f} {su:8.2f} {flops / 1e9:8.3f} {flops / (su * 1e-6) / 1e12 if su else 0:18.2f}")
    print(f"total {(steps[-1][1] - t0) / 1e3:.1f} us: step kernels {tot_step:.1f}, column kernels {tot_col:.1f}, gaps {tot_gap:.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        parse(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
