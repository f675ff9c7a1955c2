#!/usr/bin/env python
"""Headline benchmark: GP fit + acquisition cycles/sec at N=4096, d=8, 65 536 candidates (BASELINE.json).

One *cycle* (SURVEY.md 8d / BASELINE.md section 2) =
    20 x [K(X,X) assembly + Cholesky + alpha + log-marginal-likelihood + analytic gradient]
       at the fixed theta schedule: the fit's 4 restarts (GP.fit / optim.py:335-354, independent L-BFGS-B runs)
       x 5 evaluations each; every restart runs in its own host thread on its own evaluation slot of the library
       (bobe_gp_mll_submit / bobe_gp_mll_wait), as GP.fit does (--fit-concurrency 1 = one bobe_gp_mll after the
       other; --fit-mode batch = lock-step rounds through bobe_gp_mll_batch)
  + 1 x refactor at the last theta    (bobe_gp_factor)
  + 1 x sweep: posterior mean & variance of all C candidates, WIPV and WIPStd scores against the
        M = 512 integration points, argmin of both  (bobe_gp_wip_sweep)
All inputs are resident in HBM before the timed region; outputs stay in HBM (only the d+2 MLL
scalars and the two argmins cross PCIe, as they do in the BO loop).

Multi-GPU (one process per GPU, torch.distributed / RCCL): weak scaling.  Every rank holds the full
factor and runs the 20 evaluations of its own restarts (the reference shards restarts over ranks,
BOBE/pool.py:298-326), sweeps its own shard of the N_gpus x C candidate set, and the ranks exchange
(min score, global index) with one all-gather; value = cycles completed by all ranks / wall time.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD spec; = 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=["tiny", "small", "headline", "large"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-class", default="trimul", help="kernel class timed with HIP events for the roofline")
    ap.add_argument("--fit-concurrency", type=int, default=4,
                    help="restarts of the fit evaluated together per bobe_gp_mll_batch call (1 = sequential)")
    ap.add_argument("--fit-mode", default="slots", choices=["slots", "batch"],
                    help="slots: one thread + evaluation slot per restart, no barrier; batch: lock-step rounds")
    ap.add_argument("--chunk", type=int, default=0, help="candidate chunk of the sweep (0 = library default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the N>1 path on a single GPU")
    return ap.parse_args()


def cpu_baseline(X, y, cand, Z, thetas, noise, sample_c=2048):
    """Oracle ("port") timed on the host cores on a bounded sample of the same cycle; extrapolated linearly."""
    from oracle import bobe_oracle as O
    try:
        from threadpoolctl import threadpool_info
        nthreads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        nthreads = os.cpu_count() or 1
    d = X.shape[1]
    t0 = time.perf_counter()
    O.cycle_value_and_grad(X, y, np.exp(thetas[0, :d]), float(np.exp(thetas[0, d])), noise)
    t_vg = time.perf_counter() - t0
    t0 = time.perf_counter()
    gp = O.OracleGP(X, y, noise=noise, kernel="rbf", lengthscales=np.exp(thetas[-1, :d]),
                    kernel_variance=float(np.exp(thetas[-1, d])))
    t_fac = time.perf_counter() - t0
    sc = min(sample_c, cand.shape[0])
    t0 = time.perf_counter()
    O.wip_sweep(gp, cand[:sc], Z, chunk=sc)
    t_sw = time.perf_counter() - t0
    cyc = len(thetas) * t_vg + t_fac + t_sw * (cand.shape[0] / sc)
    return {"value": 1.0 / cyc, "unit": "cycles/s", "cores": int(nthreads), "kind": "port",
            "sample": f"1 of {len(thetas)} value+grad ({t_vg:.2f}s), 1 refactor ({t_fac:.2f}s), "
                      f"{sc} of {cand.shape[0]} candidates ({t_sw:.2f}s); extrapolated linearly; "
                      f"NumPy/SciPy-OpenBLAS fp64"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from bobe_amd import _lib
    from bobe_amd.gp import GP
    from bobe_amd.synthetic import CONFIGS, synthetic_problem, theta_schedule

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback)")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()      # rehearsal: several ranks may share one GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if args.backend == "nccl" else None   # where the all-gather payload lives
    from bobe_amd.dist_sweep import merge_argmin, merge_best_fit

    N, d, Cn, M = CONFIGS[args.config]
    noise = 1e-6
    X, y, cand, Z = synthetic_problem(N, d, Cn, M, noise=noise, cand_offset=rank * Cn)
    thetas = theta_schedule(d)
    gp = GP(X, y, noise=noise, kernel="rbf", lengthscales=np.full(d, 0.6), kernel_variance=1.0, device=local)
    lib, h = gp._lib, gp._h
    if args.chunk:
        _lib.check(lib.bobe_gp_set_chunk(h, args.chunk), "set_chunk")
    # inputs resident in HBM; outputs stay in HBM
    cand_d = torch.from_numpy(cand).to(dev)
    Z_d = torch.from_numpy(Z).to(dev)
    out_mean = torch.empty(Cn, dtype=torch.float64, device=dev)
    out_var = torch.empty_like(out_mean)
    out_wipv = torch.empty_like(out_mean)
    out_wipstd = torch.empty_like(out_mean)
    torch.cuda.synchronize()

    ls_last = np.ascontiguousarray(np.exp(thetas[-1, :d]))
    kv_last = float(np.exp(thetas[-1, d]))
    grad = np.empty(d + 1)
    mll = C.c_double()
    av, asd, mv, ms = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
    last = {}

    R = max(1, args.fit_concurrency)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, R))
    ls_all = np.ascontiguousarray(np.exp(thetas[:, :d]))
    kv_all = np.ascontiguousarray(np.exp(thetas[:, d]))
    mll_b = np.empty(len(thetas))
    grad_b = np.empty((len(thetas), d + 1))

    def fit_evals():
        """the 20 value+gradient evaluations; returns (best mll, its theta, last mll)"""
        best = (-np.inf, None)
        if R == 1:
            for k, th in enumerate(thetas):
                _lib.check(lib.bobe_gp_mll(h, _lib.ptr(ls_all[k]), float(kv_all[k]), C.byref(mll), _lib.ptr(grad)), "mll")
                mll_b[k] = mll.value
        elif args.fit_mode == "slots":
            def chain(r):                            # restart r: its evaluations, one after the other, on slot r
                m_, g_ = C.c_double(), np.empty(d + 1)
                for k in range(r, len(thetas), R):
                    _lib.check(lib.bobe_gp_mll_submit(h, r, _lib.ptr(ls_all[k]), float(kv_all[k]), 1), "mll_submit")
                    _lib.check(lib.bobe_gp_mll_wait(h, r, C.byref(m_), _lib.ptr(g_)), "mll_wait")
                    mll_b[k] = m_.value
                    grad_b[k] = g_
            futs = [pool.submit(chain, r) for r in range(R)]
            for f in futs:
                f.result()
        else:
            for k0 in range(0, len(thetas), R):      # round k0/R of the R restarts
                nb_ = min(R, len(thetas) - k0)
                _lib.check(lib.bobe_gp_mll_batch(h, nb_, _lib.ptr(ls_all[k0:k0 + nb_]), _lib.ptr(kv_all[k0:k0 + nb_]),
                                                 _lib.ptr(mll_b[k0:k0 + nb_]), _lib.ptr(grad_b[k0:k0 + nb_]), None),
                           "mll_batch")
        for k, th in enumerate(thetas):
            if mll_b[k] > best[0]:
                best = (float(mll_b[k]), th)
        return best

    def cycle():
        best = fit_evals()
        mll.value = float(mll_b[-1])
        _lib.check(lib.bobe_gp_set_hyper(h, _lib.ptr(ls_last), kv_last, noise), "set_hyper")
        _lib.check(lib.bobe_gp_factor(h), "factor")
        _lib.check(lib.bobe_gp_wip_sweep(h, _lib.ptr(cand_d), Cn, _lib.ptr(Z_d), M, 1.0, _lib.ptr(out_wipv),
                                         _lib.ptr(out_wipstd), _lib.ptr(out_mean), _lib.ptr(out_var),
                                         C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "sweep")
        # the path's exchange step: one all-gather of (min score, global index) — lowest global index wins
        # ties (jnp.argmin) — and one of (best mll, theta) for the restart-sharded fit (pool.py:322-326)
        gmin, gidx = merge_argmin(ms.value, rank * Cn + asd.value, device=coll_dev)
        bmll, bth = merge_best_fit(best[0], best[1], device=coll_dev)
        last.update(mll=mll.value, best_mll=float(bmll), argmin=int(gidx), min_wipstd=float(gmin))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lib.bobe_gp_sync(h)

    for _ in range(args.warmup):
        cycle()
    prof_tag = _lib.PROF[args.profile_class]
    lib.bobe_gp_profile_select(h, prof_tag)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle()
    barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, launches = C.c_double(), C.c_int64()
    lib.bobe_gp_profile_read(h, C.byref(tot_ms), C.byref(launches))
    lib.bobe_gp_profile_select(h, 0)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    fit_ms = {}
    if rank == 0:               # the fit alone, both ways (outside the timed region)
        for r_ in sorted({1, R}):
            R_keep, R = R, r_
            fit_evals()
            lib.bobe_gp_sync(h)
            t1 = time.perf_counter()
            fit_evals()
            lib.bobe_gp_sync(h)
            fit_ms["concurrency_%d" % r_] = (time.perf_counter() - t1) * 1e3
            R = R_keep
    sub_ms, lbfgs = {}, None
    if rank == 0:               # the other two sub-times of a cycle and a real L-BFGS-B fit, untimed region
        lib.bobe_gp_sync(h)
        t1 = time.perf_counter()
        _lib.check(lib.bobe_gp_set_hyper(h, _lib.ptr(ls_last), kv_last, noise), "set_hyper")
        _lib.check(lib.bobe_gp_factor(h), "factor")
        lib.bobe_gp_sync(h)
        t2 = time.perf_counter()
        _lib.check(lib.bobe_gp_wip_sweep(h, _lib.ptr(cand_d), Cn, _lib.ptr(Z_d), M, 1.0, _lib.ptr(out_wipv),
                                         _lib.ptr(out_wipstd), _lib.ptr(out_mean), _lib.ptr(out_var),
                                         C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "sweep")
        lib.bobe_gp_sync(h)
        t3 = time.perf_counter()
        sub_ms = {"fit": fit_ms.get("concurrency_%d" % R), "refactor": (t2 - t1) * 1e3, "sweep": (t3 - t2) * 1e3}
        if args.config != "tiny":
            # GP.fit as the BO loop calls it for N >= 750 (bo.py:651-653): 4 restarts (pool.py:277-286 recipe), maxiter 200
            from bobe_amd.bo import gp_fit
            calls = [0]
            orig = gp.mll_data

            def counted(*a, **k):
                calls[0] += 1
                return orig(*a, **k)
            gp.mll_data = counted
            t4 = time.perf_counter()
            r_fit = gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7))
            t5 = time.perf_counter()
            gp.mll_data = orig
            _lib.check(lib.bobe_gp_set_hyper(h, _lib.ptr(ls_last), kv_last, noise), "set_hyper")   # back to the cycle's state
            _lib.check(lib.bobe_gp_factor(h), "factor")
            lbfgs = {"restarts": 4, "maxiter": 200, "seconds": t5 - t4, "evaluations": calls[0],
                     "ms_per_evaluation": (t5 - t4) * 1e3 / max(calls[0], 1), "mll": float(r_fit["mll"])}
    if rank == 0:
        # Cholesky GF/s: mean device time of the factorisation alone (HIP events on the handle's stream)
        potrf_ms = C.c_double()
        lib.bobe_debug_time_potrf(h, 3, C.byref(potrf_ms))
        potrf_b_ms = C.c_double()
        if R > 1:
            _lib.check(lib.bobe_debug_time_potrf_batch(h, min(R, 8), 3, C.byref(potrf_b_ms)), "time_potrf_batch")
        Np = (N + 127) // 128 * 128
        chunk = args.chunk or 8192
        # k_trimul = one launch per candidate chunk: V = Linv K(X,C) (N^2 per candidate, triangular) fused with
        # the cross-covariance rows W_Z^T K(X,C) (2 N M per candidate)
        flops_per_launch = {"trimul": (float(N) * N + 2.0 * N * M) * min(chunk, Cn),
                            "syrk": None, "lauum": 2.0 * N ** 3 / 3.0}.get(args.profile_class)
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_k_%s.json" % args.profile_class)
        if os.path.exists(tf):      # HBM bytes per launch from rocprofv3 PMC passes of this same command (tools/pmc_traffic.py)
            tj = json.load(open(tf))
            if (tj.get("N"), tj.get("C"), tj.get("chunk")) == (N, Cn, chunk):
                traffic = tj["hbm_bytes_per_launch"]
        roof = None
        if flops_per_launch and launches.value:
            avg_s = tot_ms.value * 1e-3 / launches.value
            ach = flops_per_launch / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": "k_" + args.profile_class, "achieved": ach,
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "avg_launch_ms": avg_s * 1e3, "launches": int(launches.value),
                    "flops_per_launch": flops_per_launch}
        out = {
            "metric": "GP fit+acquisition cycles/sec at N=4096 d=8, 65536 cands; Cholesky GF/s",
            "value": world * args.steps / elapsed, "unit": "cycles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic RBF GP N={N} d={d}, {Cn} candidates per GPU, M={M}, fp64 "
                                   + {"headline": "(BASELINE.json configs[2])", "small": "(BASELINE.json configs[1])"}.get(
                                       args.config, "(not a BASELINE.json config)"),
                       "N": N, "d": d, "candidates_per_gpu": Cn, "M": M, "evals_per_cycle": len(thetas),
                       "fit": f"{R} restarts x {len(thetas) // R} value+gradient evaluations, restarts concurrent ({args.fit_mode})"
                              if R > 1 else f"{len(thetas)} sequential value+gradient evaluations",
                       "parallelism": f"candidate-sharded x{world}"},
            "cholesky_gflops": (N ** 3 / 3.0) / (potrf_ms.value * 1e-3) / 1e9,
            "cholesky_ms": potrf_ms.value,
            "cholesky_concurrent": ({"in_flight": min(R, 8), "ms_all": potrf_b_ms.value,
                                     "gflops": min(R, 8) * (N ** 3 / 3.0) / (potrf_b_ms.value * 1e-3) / 1e9,
                                     "frac_of_fp64_mfma_peak": min(R, 8) * (N ** 3 / 3.0) / (potrf_b_ms.value * 1e-3) / 1e12
                                     / FP64_MFMA_PEAK_TFLOPS} if R > 1 else None),
            "fit_ms": fit_ms,
            "sub_ms": sub_ms,
            "lbfgs_fit": lbfgs,
            "check": last,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(X, y, cand, Z, thetas, noise)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()              # rank 0's secondary measurements are done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
