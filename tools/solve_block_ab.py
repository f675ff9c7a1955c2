"""A/B of the ways the sweep forms V = L^-1 K(X, C) (round 6, verdict item 1):

  plain    the product with the explicit inverse factor (k_trimul, cross tiles fused)          refine_kappa < 0
  hybrid   blocked forward substitution (k_blk_step), diagonal blocks b, update panels p        refine_kappa = 0, block b

on the headline workload (N = 4096, d = 8, 65 536 candidates, M = 512; well conditioned, so both must agree to rounding) -
times only; the accuracy side is tests/test_gpu_conditioning.py.  Writes gpurun_out/r06_solve_block_ab_<config>.txt.
The committed profiles/r06_solve_block_ab_{headline,small}.txt were written by this script at commit 01a1561 + the first
form of the kernel, when round 5's refinement step (k_trimul -> k_trimul_resid -> k_trimul_add, line "refine") still
existed: 60.6 ms per sweep against 29.8 (b = 128, panels of 512 rows, 32 768 candidates per launch sequence)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bobe_amd import _lib
    from bobe_amd.gp import GP
    from bobe_amd.synthetic import CONFIGS, synthetic_problem
    cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
    N, d, Cn, M = CONFIGS[cfg]
    noise = 1e-6
    X, y, cand, Z = synthetic_problem(N, d, Cn, M, noise=noise)
    dev = torch.device("cuda", 0)
    gp = GP(X, y, noise=noise, kernel="rbf", lengthscales=np.full(d, 0.6), kernel_variance=1.0)
    lib, h = gp._lib, gp._h
    cand_d, Z_d = torch.from_numpy(cand).to(dev), torch.from_numpy(Z).to(dev)
    outs = {k: torch.empty(Cn, dtype=torch.float64, device=dev) for k in ("mean", "var", "wipv", "wipstd")}
    av, asd, mv, ms = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()

    def sweep():
        _lib.check(lib.bobe_gp_wip_sweep(h, _lib.ptr(cand_d), Cn, _lib.ptr(Z_d), M, 1.0, _lib.ptr(outs["wipv"]),
                                         _lib.ptr(outs["wipstd"]), _lib.ptr(outs["mean"]), _lib.ptr(outs["var"]),
                                         C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "sweep")

    def timed(tag_class=None, reps=3):
        sweep()
        lib.bobe_gp_sync(h)
        tms, n = C.c_double(), C.c_int64()
        per = {}
        for cls in ("trimul", "crossvv"):
            lib.bobe_gp_profile_select(h, _lib.PROF[cls])
            sweep()
            lib.bobe_gp_profile_read(h, C.byref(tms), C.byref(n))
            per[cls] = (tms.value, n.value)
        lib.bobe_gp_profile_select(h, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sweep()
        lib.bobe_gp_sync(h)
        return (time.perf_counter() - t0) / reps * 1e3, per

    rows = []
    ref = None
    variants = [("plain", -1.0, 128, 512, 0)]
    for b, pn in ((128, 128), (128, 256), (128, 512), (128, 1024), (256, 512), (256, 1024), (512, 512), (1024, 1024)):
        for ch in (0, 16384, 32768):
            variants.append((f"hybrid b={b} panel={pn} chunk={ch or 8192}", 0.0, b, pn, ch))
    for name, kappa, block, la, ch in variants:
        gp.refine_kappa = kappa
        gp.solve_block = block
        _lib.check(lib.bobe_debug_solve_opts(h, la, ch), "solve_opts")
        gp.recompute_cholesky()
        ms_sweep, per = timed()
        res = {k: v.cpu().numpy().copy() for k, v in outs.items()}
        if ref is None:
            ref = res
        dev_ = {k: float(np.max(np.abs(res[k] - ref[k])) / np.max(np.abs(ref[k]))) for k in res}
        rows.append((name, ms_sweep, per, dev_, int(asd.value)))
        print(name, f"{ms_sweep:.2f} ms", per, dev_, asd.value, flush=True)
    out = os.path.join("gpurun_out", f"r06_solve_block_ab_{cfg}.txt")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(out, "w") as fh:
        fh.write(f"# tools/solve_block_ab.py {cfg}: N={N} d={d} C={Cn} M={M}, noise {noise}; sweep wall ms (mean of 3), the solve's\n"
                 "# launches between HIP events (class trimul: every launch of solve_v per chunk), the separate cross launches,\n"
                 "# max |delta| / max |plain| of the four outputs, the chosen candidate\n")
        for name, ms_sweep, per, dev_, am in rows:
            fh.write(f"{name:<34} sweep {ms_sweep:8.2f} ms   solve {per['trimul'][0]:8.2f} ms / {per['trimul'][1]:3d}   "
                     f"cross {per['crossvv'][0]:7.2f} ms / {per['crossvv'][1]:3d}   "
                     + " ".join(f"{k} {v:.1e}" for k, v in dev_.items()) + f"   argmin {am}\n")


if __name__ == "__main__":
    main()
