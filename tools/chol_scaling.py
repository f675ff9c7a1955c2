"""Factorisation timing vs N on the GPU box (device time via HIP events inside the library, bobe_debug_time_potrf*).

  python tools/chol_scaling.py [N ...]          alone, four in lock step, one value+gradient evaluation
  python tools/chol_scaling.py fill [N ...]     the update fillers of the panel launches (gp_factor.hip: fill_pays):
                                                a lone factorisation with BOBE_FILL=0 (off), =2 (forced on) and =1 (the
                                                rule), each in its own process (the switch is read once per process)
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(N, lockstep=True, reps=5):
    from bobe_amd import _lib
    from bobe_amd.gp import GP
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(N, 8))
    gp = GP(X, np.sin(X.sum(1)), noise=1e-4, lengthscales=np.full(8, 0.6))
    ms = C.c_double()
    _lib.check(gp._lib.bobe_debug_time_potrf(gp._h, reps, C.byref(ms)), "time_potrf")
    out = {"N": N, "alone_ms": ms.value}
    if lockstep:
        _lib.check(gp._lib.bobe_debug_time_potrf_lockstep(gp._h, 4, max(2, reps // 2), C.byref(ms)), "lockstep")
        out["lockstep4_ms"] = ms.value
        ls = np.full(8, 0.55)
        gp.mll_data(ls, 1.0)
        t0 = time.perf_counter()
        for _ in range(3):
            gp.mll_data(ls, 1.0)
        out["value_grad_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "one":                  # child of the `fill` mode
        print(json.dumps(measure(int(args[1]), lockstep=False, reps=10)))
    elif args and args[0] == "fill":
        sizes = [int(a) for a in args[1:]] or [2048, 2560, 3072, 4096, 4608, 4992, 5120, 6144, 8192]
        print("# lone factorisation, device ms (mean of 10 after one untimed pass); rule = nb >= 28 and the first panel launch")
        print("# occupies <= 35 % of the CUs (fill_pays, gp_factor.hip)")
        print("#    N   nb   fillers off   forced on   default rule   rule says")
        for N in sizes:
            row = {}
            for mode in ("0", "2", "1"):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(N)], env=dict(os.environ, BOBE_FILL=mode),
                                   capture_output=True, text=True, timeout=600)
                row[mode] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["alone_ms"]
            on = abs(row["1"] - row["2"]) < abs(row["1"] - row["0"])
            print(f"{N:6d} {-(-N // 128):4d}   {row['0']:9.3f}   {row['2']:9.3f}   {row['1']:10.3f}      {'on' if on else 'off'}"
                  f"   (forced on / off = {row['2'] / row['0']:.3f})", flush=True)
    else:
        for N in ([int(a) for a in args] or (512, 1024, 2048, 4096, 8192, 12288)):
            r = measure(N)
            tf = lambda ms, b=1: b * N ** 3 / 3 / ms / 1e9                     # noqa: E731
            print(f"N={N:6d}  alone {r['alone_ms']:8.3f} ms = {tf(r['alone_ms']):6.2f} TF/s ({tf(r['alone_ms']) / 78.6 * 100:4.1f} %)"
                  f"  x4 lock step {r['lockstep4_ms']:8.3f} ms = {tf(r['lockstep4_ms'], 4):6.2f} TF/s ({tf(r['lockstep4_ms'], 4) / 78.6 * 100:4.1f} %)"
                  f"  value+grad {r['value_grad_ms']:8.3f} ms = {N ** 3 / r['value_grad_ms'] / 1e9:6.2f} TF/s", flush=True)
