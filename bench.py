#!/usr/bin/env python
"""Headline benchmark: GP fit + acquisition cycles/sec at N=4096, d=8, 65 536 candidates (BASELINE.json).

One *cycle* (SURVEY.md 8d / BASELINE.md section 2) =
    20 x [K(X,X) assembly + Cholesky + alpha + log-marginal-likelihood + analytic gradient]
       at the fixed theta schedule: the fit's 4 restarts (GP.fit / optim.py:335-354, independent L-BFGS-B runs)
       x 5 evaluations each.  --fit-mode batch (default): lock-step rounds through bobe_gp_mll_batch (one batched launch
       sequence per round: what GP.fit does); --fit-mode slots: every restart in its own host thread on its own
       evaluation slot (bobe_gp_mll_submit / bobe_gp_mll_wait); --fit-concurrency 1: one after the other.
  + 1 x refactor at the last theta    (bobe_gp_factor)
  + 1 x sweep: posterior mean & variance of all C candidates, WIPV and WIPStd scores against the
        M = 512 integration points, argmin of both  (bobe_gp_wip_sweep)
All inputs are resident in HBM before the timed region; outputs stay in HBM (only the d+2 MLL
scalars and the two argmins cross PCIe, as they do in the BO loop).

Multi-GPU (one process per GPU, torch.distributed / RCCL).  ``--gpus N`` without a torchrun environment starts the
N ranks itself (a child ``python -m torch.distributed.run``, before this process touches the GPU) and exits with its
status; under torchrun it insists that WORLD_SIZE == N.
  weak  (--config headline, default): every rank holds the full factor, runs the 20 evaluations of its own restarts
        and sweeps its own 65 536 candidates (rows [rank*C, (rank+1)*C) of the Sobol set); value = cycles of all
        ranks / wall time.
  strong (--config shard = BASELINE.json configs[3]): ONE cycle over 262 144 candidates split into contiguous shards
        (np.array_split bounds), the fit's 4 restarts split over the ranks the way the reference's MPI pool splits
        them (BOBE/pool.py:298-326); value = cycles / wall time.
Either way the ranks exchange (min score, global index) with one all-gather per sweep and (best mll, theta) with one
per fit: through torch.distributed (default) or, with ``--exchange rccl``, through the C ABI's own entry points
(bobe_mgpu_wip_sweep = shard sweep + ncclAllGather + merge in one call, bobe_mgpu_best_fit).
With N > 1 in the weak mode the line also carries a ``shard`` sub-record: after the weak timed region the same ranks time
the STRONG config-4 cycle (262 144 candidates and the fit's restarts split N ways) - what the scaling curve of the weak
mode, N independent cycles, cannot say.  ``--no-secondary``: timed cycles only (for kernel traces that read per cycle).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD spec; = 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz)
SHARD_TOTAL_CANDIDATES = 262144


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=["tiny", "small", "headline", "large", "shard"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the timed cycles: no phase timers, Cholesky timers, L-BFGS fit or CPU baseline afterwards "
                         "(a rocprofv3 kernel trace of such a run reads per cycle)")
    ap.add_argument("--no-shard-record", action="store_true", help="N > 1, weak mode: skip the strong config-4 sub-record")
    ap.add_argument("--profile-class", default="trimul", help="kernel class timed with HIP events for the roofline")
    ap.add_argument("--fit-concurrency", type=int, default=4, help="restarts of the fit in flight together (1 = sequential)")
    ap.add_argument("--fit-mode", default="batch", choices=["slots", "batch"],
                    help="batch: lock-step rounds through bobe_gp_mll_batch (what GP.fit does); slots: one thread + "
                         "evaluation slot per restart, no barrier")
    ap.add_argument("--chunk", type=int, default=0, help="candidate chunk of the sweep (0 = library default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the N>1 path on a single GPU")
    ap.add_argument("--exchange", default="torch", choices=["torch", "rccl"],
                    help="who issues the all-gathers: torch.distributed (default) or the library's own RCCL communicator "
                         "(bobe_mgpu_*, include/bobe_gp.h)")
    ap.add_argument("--shard-candidates", type=int, default=SHARD_TOTAL_CANDIDATES,
                    help="total candidates of --config shard (tests use a smaller set)")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(args) -> int:
    """--gpus N from a plain shell: start the N ranks as a child torchrun.  Runs BEFORE torch / HIP are touched in
    this process (a process that has initialised the GPU must not be replaced or forked on this pool)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(X, y, cand, Z, thetas, noise, gpu_check, sample_c=16384, samples=5):
    """The cycle on the host cores (oracle/cpu_port.py, kind "port": the same rank-1 algorithm on torch-CPU fp64 ->
    LAPACK/BLAS, all threads): one warm-up, then the MEDIAN of ``samples`` value+gradient evaluations, one refactor,
    one sweep of a ``sample_c``-candidate sample; extrapolated linearly to the full cycle."""
    from oracle import cpu_port as P
    import torch
    d = X.shape[1]
    ls0, kv0 = np.exp(thetas[0, :d]), float(np.exp(thetas[0, d]))
    # thread count: the fastest of a few candidates on one evaluation each (all logical CPUs is NOT the fastest on a
    # box whose CPU share is smaller than its core count: 6.1 s with 128 threads against 1.4 s with 16 at N = 4096)
    ncpu = os.cpu_count() or 1
    tried = {}
    for nt in sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu), torch.get_num_threads()}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        P.cycle_value_and_grad(X, y, ls0, kv0, noise)
        tried[nt] = time.perf_counter() - t0
    torch.set_num_threads(min(tried, key=tried.get))
    mll0, g0 = P.cycle_value_and_grad(X, y, ls0, kv0, noise)              # warm-up (also the parity reference)
    t_vg = []
    for _ in range(samples):
        t0 = time.perf_counter()
        P.cycle_value_and_grad(X, y, ls0, kv0, noise)
        t_vg.append(time.perf_counter() - t0)
    ls1, kv1 = np.exp(thetas[-1, :d]), float(np.exp(thetas[-1, d]))
    P.factor(X, y, ls1, kv1, noise)
    t0 = time.perf_counter()
    f = P.factor(X, y, ls1, kv1, noise)
    t_fac = time.perf_counter() - t0
    sc = min(sample_c, cand.shape[0])
    P.wip_sweep(f, cand[:min(sc, 512)], Z, chunk=512)
    t0 = time.perf_counter()
    sw = P.wip_sweep(f, cand[:sc], Z, chunk=2048)
    t_sw = time.perf_counter() - t0
    vg = float(np.median(t_vg))
    cyc = len(thetas) * vg + t_fac + t_sw * (cand.shape[0] / sc)
    host = P.host_description()
    par = {}
    if gpu_check is not None:
        par = {"rel_dmll_theta0": abs(gpu_check["mll0"] - mll0) / abs(mll0),
               "rel_dgrad_theta0": float(np.max(np.abs(gpu_check["grad0"] - g0)) / np.max(np.abs(g0))),
               "sample_wipstd_max_rel": float(np.max(np.abs(gpu_check["wipstd"][:sc] - sw["wipstd"]) / np.abs(sw["wipstd"]))),
               "sample_argmin_equal": bool(int(np.argmin(gpu_check["wipstd"][:sc])) == sw["argmin_s"])}
    return {"value": 1.0 / cyc, "unit": "cycles/s", "cores": host["threads"], "kind": "port",
            "sample": f"median of {samples} value+grad after 1 warm-up ({vg:.3f}s; min {min(t_vg):.3f}, max {max(t_vg):.3f}) "
                      f"x {len(thetas)}, 1 refactor ({t_fac:.3f}s), {sc} of {cand.shape[0]} candidates ({t_sw:.3f}s) "
                      f"extrapolated linearly; same rank-1 sweep algorithm as the GPU",
            "host": host, "seconds_per_cycle": cyc, "thread_calibration_s": {str(k): round(v, 3) for k, v in tried.items()},
            "parity_vs_gpu": par}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}` "
                         f"or plain `python bench.py --gpus {args.gpus}`")
    import torch
    import torch.distributed as dist
    from bobe_amd import _lib
    from bobe_amd.gp import GP
    from bobe_amd.synthetic import CONFIGS, synthetic_problem, theta_schedule

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
        assert dist.get_world_size() == args.gpus and dist.get_backend() == args.backend
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if args.backend == "nccl" else None   # where the all-gather payload lives
    from bobe_amd.dist_sweep import merge_argmin, merge_best_fit, shard_bounds
    use_rccl = args.exchange == "rccl"                 # the C ABI's exchange step instead of torch's collectives
    if use_rccl:
        from bobe_amd import mgpu
        mgpu.init_from_torch(local)

    strong = args.config == "shard"
    from bobe_amd.synthetic import sobol_candidates
    if strong:
        N, d, _, M = CONFIGS["headline"]
        c_total = int(args.shard_candidates)
        c_lo, c_hi = shard_bounds(c_total, world, rank)
    else:
        N, d, c_per, M = CONFIGS[args.config]
        c_total = world * c_per
        c_lo, c_hi = rank * c_per, (rank + 1) * c_per
    Cn = c_hi - c_lo
    noise = 1e-6
    X, y, cand, Z = synthetic_problem(N, d, Cn, M, noise=noise, cand_offset=c_lo)
    thetas = theta_schedule(d)
    gp = GP(X, y, noise=noise, kernel="rbf", lengthscales=np.full(d, 0.6), kernel_variance=1.0, device=local)
    lib, h = gp._lib, gp._h
    if args.chunk:
        _lib.check(lib.bobe_gp_set_chunk(h, args.chunk), "set_chunk")
    Z_d = torch.from_numpy(Z).to(dev)

    def make_work(cand_np, lo, total):
        """a candidate set resident in HBM with its output vectors (which stay in HBM)"""
        n = cand_np.shape[0]
        w = {"cand": torch.from_numpy(cand_np).to(dev), "n": n, "lo": lo, "total": total}
        for k in ("mean", "var", "wipv", "wipstd"):
            w[k] = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
        return w
    work = make_work(cand, c_lo, c_total)
    Cn = work["n"]
    out_wipstd = work["wipstd"]
    torch.cuda.synchronize()

    ls_last = np.ascontiguousarray(np.exp(thetas[-1, :d]))
    kv_last = float(np.exp(thetas[-1, d]))
    grad = np.empty(d + 1)
    mll = C.c_double()
    av, asd, mv, ms = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
    last = {}

    R_total = max(1, args.fit_concurrency)              # restarts of the fit (= evaluations in flight on one GPU)
    # restarts this rank runs: all of them (weak: its own fit), or its np.array_split share (strong: pool.py:298-326)
    restart_share = list(range(*shard_bounds(R_total, world, rank)))
    my_restarts = list(range(R_total)) if not strong else restart_share
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, R_total))
    ls_all = np.ascontiguousarray(np.exp(thetas[:, :d]))
    kv_all = np.ascontiguousarray(np.exp(thetas[:, d]))
    mll_b = np.full(len(thetas), -np.inf)
    grad_b = np.empty((len(thetas), d + 1))

    def fit_evals(restarts, mode, R=None):
        """the value+gradient evaluations of ``restarts`` (restart r owns thetas r, r+R, r+2R, ...; R = R_total)"""
        R = R_total if R is None else R
        ks = [[k for k in range(r, len(thetas), R)] for r in restarts]
        if not ks:
            return (-np.inf, thetas[0])
        if len(restarts) == 1 or mode == "sequential":
            for chain_k in ks:
                for k in chain_k:
                    _lib.check(lib.bobe_gp_mll(h, _lib.ptr(ls_all[k]), float(kv_all[k]), C.byref(mll), _lib.ptr(grad)), "mll")
                    mll_b[k] = mll.value
                    grad_b[k] = grad
        elif mode == "slots":
            def chain(slot, chain_k):                # a restart: its evaluations one after the other on its slot
                m_, g_ = C.c_double(), np.empty(d + 1)
                for k in chain_k:
                    _lib.check(lib.bobe_gp_mll_submit(h, slot, _lib.ptr(ls_all[k]), float(kv_all[k]), 1), "mll_submit")
                    _lib.check(lib.bobe_gp_mll_wait(h, slot, C.byref(m_), _lib.ptr(g_)), "mll_wait")
                    mll_b[k] = m_.value
                    grad_b[k] = g_
            for f in [pool.submit(chain, s, ck) for s, ck in enumerate(ks)]:
                f.result()
        else:                                        # lock-step rounds: evaluation j of every restart together
            for j in range(max(len(ck) for ck in ks)):
                idx = np.array([ck[j] for ck in ks if j < len(ck)])
                lsr, kvr = np.ascontiguousarray(ls_all[idx]), np.ascontiguousarray(kv_all[idx])
                mr, gr = np.empty(len(idx)), np.empty((len(idx), d + 1))
                _lib.check(lib.bobe_gp_mll_batch(h, len(idx), _lib.ptr(lsr), _lib.ptr(kvr), _lib.ptr(mr), _lib.ptr(gr), None),
                           "mll_batch")
                mll_b[idx] = mr
                grad_b[idx] = gr
        mine = [k for ck in ks for k in ck if np.isfinite(mll_b[k])]      # a NaN (K not positive definite) never wins
        if not mine:
            return (-np.inf, thetas[0])
        kb = max(mine, key=lambda k: mll_b[k])
        return (float(mll_b[kb]), thetas[kb])

    fit_mode = "sequential" if R_total == 1 else args.fit_mode

    def local_sweep(w):
        _lib.check(lib.bobe_gp_wip_sweep(h, _lib.ptr(w["cand"]), w["n"], _lib.ptr(Z_d), M, 1.0, _lib.ptr(w["wipv"]),
                                         _lib.ptr(w["wipstd"]), _lib.ptr(w["mean"]), _lib.ptr(w["var"]),
                                         C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "sweep")

    def cycle(w=None, restarts=None, phase=None):
        """one cycle on candidate set ``w`` with the fit's ``restarts`` on this rank; ``phase`` collects wall times"""
        w = work if w is None else w
        restarts = my_restarts if restarts is None else restarts
        t_0 = time.perf_counter()
        best = fit_evals(restarts, fit_mode if len(restarts) > 1 else "sequential")
        t_1 = time.perf_counter()
        _lib.check(lib.bobe_gp_set_hyper(h, _lib.ptr(ls_last), kv_last, noise), "set_hyper")
        _lib.check(lib.bobe_gp_factor(h), "factor")
        t_2 = time.perf_counter()
        # the path's exchange step: one all-gather of (min score, global index) — lowest global index wins
        # ties (jnp.argmin) — and one of (best mll, theta) for the restart-sharded fit (pool.py:322-326)
        if use_rccl:
            # the shipped entry point: shard sweep + ncclAllGather + merge inside the library
            _lib.check(lib.bobe_mgpu_wip_sweep(h, _lib.ptr(w["cand"]) if w["n"] else None, w["n"], w["lo"], _lib.ptr(Z_d), M, 1.0,
                                               _lib.ptr(w["wipv"]), _lib.ptr(w["wipstd"]), _lib.ptr(w["mean"]),
                                               _lib.ptr(w["var"]), C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)),
                       "mgpu_wip_sweep")
            t_3 = time.perf_counter()
            gmin, gidx = ms.value, asd.value
            bmll, bth = mgpu.best_fit(best[0], best[1])
        else:
            if w["n"]:
                local_sweep(w)
                loc = (ms.value, w["lo"] + asd.value)
            else:
                loc = (float("inf"), 2 ** 52)                          # a rank without candidates never wins
            t_3 = time.perf_counter()
            gmin, gidx = merge_argmin(loc[0], loc[1], device=coll_dev)
            bmll, bth = merge_best_fit(best[0], best[1], device=coll_dev)
        t_4 = time.perf_counter()
        if phase is not None:
            for k_, v_ in (("fit", t_1 - t_0), ("refactor", t_2 - t_1), ("sweep", t_3 - t_2), ("exchange", t_4 - t_3)):
                phase[k_] = phase.get(k_, 0.0) + v_
        if os.environ.get("BENCH_DEBUG"):
            print(f"[rank {rank}] restarts {restarts} local best {best[0]!r} merged {bmll!r} mll_b {np.round(mll_b, 3).tolist()}",
                  file=sys.stderr, flush=True)
        last.update(best_mll=float(bmll), argmin=int(gidx), min_wipstd=float(gmin))

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

    # ---- N > 1, weak mode: the same ranks now time the STRONG config-4 cycle (candidates and restarts split N ways)
    shard_rec = None
    weak_check = dict(last)
    if world > 1 and not strong and not args.no_shard_record:
        s_total = int(args.shard_candidates)
        s_lo, s_hi = shard_bounds(s_total, world, rank)
        swork = make_work(sobol_candidates(d, s_hi - s_lo, s_lo), s_lo, s_total)
        cycle(swork, restart_share)                                        # untimed: workspace growth, first touch
        barrier()
        ph = {}
        t1 = time.perf_counter()
        for _ in range(args.steps):
            cycle(swork, restart_share, ph)
        barrier()
        s_el = time.perf_counter() - t1
        vals = torch.tensor([s_el, ph["fit"], ph["refactor"], ph["sweep"], ph["exchange"]], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)                        # the slowest rank sets each figure
        vals = (vals / args.steps * 1e3).tolist()
        shard_rec = {"workload": f"N={N} d={d}, {s_total} candidates split over {world} GPUs, the fit's {R_total} restarts "
                                 f"split over the ranks (BASELINE.json configs[3])",
                     "scaling": "strong", "cycles_per_s": 1e3 / vals[0], "ms_per_cycle": vals[0], "fit_ms": vals[1],
                     "refactor_ms": vals[2], "sweep_ms": vals[3], "exchange_ms": vals[4], "candidates_total": s_total,
                     "candidates_per_gpu": s_hi - s_lo, "restarts_rank0": restart_share, "steps": args.steps,
                     "check": dict(last),
                     "note": "max over ranks per phase; exchange_ms = the two all-gathers INCLUDING the wait for the slowest rank "
                             "(a rank that finished its fit / sweep early waits there); with --exchange rccl the sweep's "
                             "all-gather is inside sweep_ms (bobe_mgpu_wip_sweep) and exchange_ms is bobe_mgpu_best_fit alone"}
        del swork

    def timed(fn, reps=5):
        """median wall time (ms) of ``reps`` calls after one untimed call (a single slow call - a first-use allocation, a
        host hiccup - would otherwise move a two-call mean by several per cent)"""
        fn()
        lib.bobe_gp_sync(h)
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            fn()
            lib.bobe_gp_sync(h)
            ts.append((time.perf_counter() - t1) * 1e3)
        return float(np.median(ts))

    secondary = rank == 0 and not args.no_secondary
    fit_ms, sub_ms, lbfgs, gpu_check = {}, {}, None, None
    if secondary:               # secondary measurements, outside the timed region: the three phases of a cycle on rank 0
        all_r = list(range(R_total))
        fit_ms["sequential"] = timed(lambda: fit_evals(all_r, "sequential"), 3)
        if R_total > 1:
            fit_ms[f"slots_{R_total}"] = timed(lambda: fit_evals(all_r, "slots"))
            fit_ms[f"lockstep_{R_total}"] = timed(lambda: fit_evals(all_r, "batch"))
            if R_total != 8:            # the same 20 evaluations as eight restarts: rounds of 8, 8 and 4 in lock step
                fit_ms["lockstep_8"] = timed(lambda: fit_evals(list(range(8)), "batch", 8))
        if strong and world > 1:
            fit_ms["this_rank_share"] = timed(lambda: fit_evals(my_restarts, fit_mode))

        def refactor():
            _lib.check(lib.bobe_gp_set_hyper(h, _lib.ptr(ls_last), kv_last, noise), "set_hyper")
            _lib.check(lib.bobe_gp_factor(h), "factor")

        key = "sequential" if R_total == 1 else (f"slots_{R_total}" if args.fit_mode == "slots" else f"lockstep_{R_total}")
        sub_ms = {"fit": fit_ms[key], "refactor": timed(refactor), "sweep": timed(lambda: local_sweep(work))}
        # kernel classes beside the dominant one, HIP events on the handle's stream over one cycle each: the two assembly
        # kernels (HBM-bound in principle: roofline_assembly) and the sweep's cross-covariance GEMM
        class_ms = {}
        for cls in ("kxc", "kxx", "crossvv"):
            lib.bobe_gp_profile_select(h, _lib.PROF[cls])
            fit_evals(all_r, fit_mode)                        # (rank 0 alone: no collective in this loop)
            refactor()
            local_sweep(work)
            t_ms, n_l = C.c_double(), C.c_int64()
            lib.bobe_gp_profile_read(h, C.byref(t_ms), C.byref(n_l))
            class_ms[cls] = (t_ms.value, int(n_l.value))
        lib.bobe_gp_profile_select(h, 0)
        gpu_check = {"mll0": float(mll_b[0]), "grad0": grad_b[0].copy(), "wipstd": out_wipstd.cpu().numpy()}
        if args.config not in ("tiny",):
            # GP.fit as the BO loop calls it for N >= 750 (bo.py:651-653): 4 restarts (pool.py:277-286 recipe), maxiter 200
            from bobe_amd.bo import gp_fit
            calls = [0]
            orig, orig_b = gp.mll_data, gp.mll_data_batch

            def counted(*a, **k):
                calls[0] += 1
                return orig(*a, **k)

            def counted_b(ls_, kv_, *a, **k):                  # (GP.fit advances its restarts in lock step: batched calls)
                calls[0] += len(kv_)
                return orig_b(ls_, kv_, *a, **k)
            gp.mll_data, gp.mll_data_batch = counted, counted_b
            t4 = time.perf_counter()
            r_fit = gp_fit(gp, maxiters=200, n_restarts=4, rng=np.random.default_rng(7), distributed=False)
            t5 = time.perf_counter()
            gp.mll_data, gp.mll_data_batch = orig, orig_b
            refactor()                                   # back to the cycle's state
            import scipy
            from bobe_amd.optim import lbfgs_driver
            lbfgs = {"restarts": 4, "maxiter": 200, "seconds": t5 - t4, "evaluations": calls[0],
                     "ms_per_evaluation": (t5 - t4) * 1e3 / max(calls[0], 1), "mll": float(r_fit["mll"]),
                     # which driver advanced the restarts: "stepped" (SciPy's reverse-communication L-BFGS-B stepped by one
                     # thread, lock-step batches) or "threads" (one scipy.optimize.minimize per restart)
                     "driver": lbfgs_driver(), "scipy": scipy.__version__}
    accurate, ref_noise, matern = None, None, None
    if secondary and world == 1 and args.config in ("headline", "small", "large"):
        try:
            # ---- (a) the sweep as the reference's regime takes it: v = L^-1 k SOLVED for (blocked forward substitution,
            # k_blk_step) instead of multiplied out with the inverse factor.  Forced on here (kappa = 0) on the headline workload;
            # at the reference's default noise it switches on by itself: (b).
            def class_time(cls, fn):
                lib.bobe_gp_profile_select(h, _lib.PROF[cls])
                fn()
                t_ms, n_l = C.c_double(), C.c_int64()
                lib.bobe_gp_profile_read(h, C.byref(t_ms), C.byref(n_l))
                lib.bobe_gp_profile_select(h, 0)
                return t_ms.value, int(n_l.value)

            _lib.check(lib.bobe_gp_set_refine_kappa(h, 0.0), "set_refine_kappa")
            refactor()
            forced_on = bool(gp.refining)
            acc_ms = timed(lambda: local_sweep(work))
            acc_w = out_wipstd.cpu().numpy().copy()
            acc_pick = int(asd.value)
            t_sv, n_sv = class_time("trimul", lambda: local_sweep(work))      # HIP events around every solve_v call (all its launches)
            t_xv, n_xv = class_time("crossvv", lambda: local_sweep(work))
            _lib.check(lib.bobe_gp_set_refine_kappa(h, 1e6), "set_refine_kappa")
            refactor()
            local_sweep(work)
            plain_w = out_wipstd.cpu().numpy()
            f_sv = float(N) * N * Cn / max(n_sv, 1)                          # N^2 flops per candidate (the triangular count)
            accurate = {"what": "bobe_gp_wip_sweep with the factor treated as ill conditioned (bobe_gp_set_refine_kappa(h, 0)): "
                                "V = L^-1 K(X,C) by blocked forward substitution (k_blk_step), the reference's solve_triangular "
                                "(gp.py:462, 571); the shipped rule switches it on where (kvar + noise) / smallest pivot > 1e6",
                        "sweep_ms": acc_ms, "sweep_ms_plain_product": sub_ms["sweep"], "ratio": acc_ms / sub_ms["sweep"],
                        "substitution_on": forced_on, "block_rows": int(lib.bobe_gp_get_solve_block(h)),
                        "roofline_solve": {"bound": "mfma", "kernel": "k_blk_step launch sequence of one candidate chunk",
                                           "achieved": f_sv / (t_sv * 1e-3 / max(n_sv, 1)) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                                           "unit": "TFLOP/s", "frac": f_sv / (t_sv * 1e-3 / max(n_sv, 1)) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                           "flops_per_chunk": f_sv, "avg_chunk_ms": t_sv / max(n_sv, 1), "chunks": n_sv},
                        "cross_ms": t_xv, "cross_launches": n_xv,
                        "wipstd_max_rel_vs_plain_product": float(np.max(np.abs(acc_w - plain_w) / np.abs(plain_w))),
                        "same_pick": bool(acc_pick == int(asd.value))}

            def whole_cycle(g, ls_ref=None):
                """fit (lock-step rounds of four) + refactor (at ls_ref, else the schedule's last theta) + sweep on another GP
                object's handle"""
                ls_ref = ls_last if ls_ref is None else ls_ref
                hh = g._h
                for j in range(len(thetas) // 4):
                    idx = np.arange(4 * j, 4 * j + 4)
                    lsr, kvr = np.ascontiguousarray(ls_all[idx]), np.ascontiguousarray(kv_all[idx])
                    mr, gr = np.empty(4), np.empty((4, d + 1))
                    _lib.check(lib.bobe_gp_mll_batch(hh, 4, _lib.ptr(lsr), _lib.ptr(kvr), _lib.ptr(mr), _lib.ptr(gr), None), "mll_batch")
                t_a = time.perf_counter()
                _lib.check(lib.bobe_gp_set_hyper(hh, _lib.ptr(ls_ref), kv_last, float(g.noise)), "set_hyper")
                st = _lib.check(lib.bobe_gp_factor(hh), "factor")
                t_b = time.perf_counter()
                _lib.check(lib.bobe_gp_wip_sweep(hh, _lib.ptr(work["cand"]), Cn, _lib.ptr(Z_d), M, 1.0, _lib.ptr(work["wipv"]),
                                                 _lib.ptr(work["wipstd"]), _lib.ptr(work["mean"]), _lib.ptr(work["var"]),
                                                 C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "sweep")
                return st, mr, (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3

            def time_cycles(g, reps=3, ls_ref=None):
                whole_cycle(g, ls_ref)
                ts, last_ = [], None
                for _ in range(reps):
                    t_a = time.perf_counter()
                    last_ = whole_cycle(g, ls_ref)
                    ts.append((time.perf_counter() - t_a) * 1e3)
                return float(np.median(ts)), last_

            # ---- (b) one cycle at the reference's default noise of 1e-8 (gp.py:201): nothing is forced, the library decides
            g8 = GP(X, y, noise=1e-8, kernel="rbf", lengthscales=np.full(d, 0.6), kernel_variance=1.0, device=local)
            ms8, (st8, mll8, rf8, sw8) = time_cycles(g8)
            ref_noise = {"noise": 1e-8, "ms_per_cycle": ms8, "cycles_per_s": 1e3 / ms8, "refactor_ms": rf8, "sweep_ms": sw8,
                         "substitution_on": bool(g8.refining), "factor_status": int(st8),
                         "finite_mll_of_last_round": int(np.sum(np.isfinite(mll8))), "pivot_floor_ulp": g8.pivot_floor_ulp,
                         "note": "same data and theta schedule; the accurate solve switches on by itself when (kvar + noise) / "
                                 "smallest pivot of the installed factor exceeds 1e6"}
            # ... and with the refactor at a longer length scale (1.2 in every dimension: what a fit of a smooth likelihood reaches),
            # where the factor's pivots fall to the noise and the rule trips: the sweep is then the substitution's, unforced
            ls_long = np.full(d, 1.2)
            ms8l, (st8l, _, rf8l, sw8l) = time_cycles(g8, ls_ref=ls_long)
            ref_noise["long_lengthscale"] = {"lengthscale": 1.2, "ms_per_cycle": ms8l, "cycles_per_s": 1e3 / ms8l, "refactor_ms": rf8l,
                                             "sweep_ms": sw8l, "substitution_on": bool(g8.refining), "factor_status": int(st8l)}
            del g8
            # ---- (c) the same cycle with the Matern-5/2 kernel (north_star names both kernels)
            gm = GP(X, y, noise=noise, kernel="matern", lengthscales=np.full(d, 0.6), kernel_variance=1.0, device=local)
            msm, (stm, mllm, rfm, swm) = time_cycles(gm)
            matern = {"kernel": "matern-5/2", "ms_per_cycle": msm, "cycles_per_s": 1e3 / msm, "refactor_ms": rfm, "sweep_ms": swm,
                      "substitution_on": bool(gm.refining), "factor_status": int(stm)}
            del gm
            refactor()
        except Exception as e:      # (a failing secondary must not cost the headline line)
            print(f"bench.py: secondary measurements (accurate sweep / noise 1e-8 / Matern) failed: {e!r}", file=sys.stderr, flush=True)
            try:
                _lib.check(lib.bobe_gp_set_refine_kappa(h, 1e6), "set_refine_kappa")
                refactor()
            except Exception:
                pass
    if rank == 0:
        chunk = args.chunk or 8192
        # k_trimul = one launch per candidate chunk: V = Linv K(X,C) (N^2 flops per candidate: the triangular count) with the
        # column sums of squares in the epilogue, plus - from the second chunk of a sweep on - the cross-covariance tiles
        # V_Z^T V of the PREVIOUS chunk (2 N M per candidate); the last chunk's cross tiles are a launch of their own
        # (k_cross_vv).  Average over a sweep's launches:
        n_ch = -(-Cn // chunk)
        flops_per_launch = {"trimul": (float(N) * N + (2.0 * N * M * (n_ch - 1) / n_ch if n_ch > 1 else 0.0)) * min(chunk, Cn),
                            "crossvv": 2.0 * N * M * min(chunk, Cn),
                            "syrk": None, "lauum": 2.0 * N ** 3 / 3.0}.get(args.profile_class)
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, "profiles", "traffic_k_%s.json" % args.profile_class)
        if os.path.exists(tf):      # HBM bytes per launch from rocprofv3 PMC passes of this same command (tools/pmc_traffic.py)
            tj = json.load(open(tf))
            if (tj.get("N"), tj.get("C"), tj.get("chunk")) == (N, Cn, chunk):
                traffic = tj["hbm_bytes_per_launch"]
                # NOT measured by this run: PMC counters need their own rocprofv3 passes (separate --pmc runs of this command)
                traffic_source = ("profiles/traffic_k_%s.json: rocprofv3 --pmc passes of `%s` (%s), collected by "
                                  "tools/pmc_traffic.py; not re-measured in this run" %
                                  (args.profile_class, tj.get("command", "python bench.py --no-secondary"),
                                   tj.get("collected", "round 3")))
        roof = None
        if flops_per_launch and launches.value:
            avg_s = tot_ms.value * 1e-3 / launches.value
            ach = flops_per_launch / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": "k_" + args.profile_class, "achieved": ach,
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": avg_s * 1e3,
                    "launches": int(launches.value),
                    "flops_per_launch": flops_per_launch,
                    "note": "largest single kernel of the cycle by time (the sweep's GEMM); the fit phase is priced in roofline_fit"}
        roof_asm, roof_cross = None, None
        if secondary:
            Np_ = (N + 127) // 128 * 128
            nb_ = Np_ // 128
            t_kxc, n_kxc = class_ms["kxc"]
            t_kxx, n_kxx = class_ms["kxx"]
            # algorithmic bytes = what an assembly must write: 8 N chunk for a K(X,C) chunk; for K(X,X) the lower
            # 128-tiles the factorisation reads, 8 * 128^2 * nb (nb + 1) / 2 per matrix (a lock-step launch assembles B of them)
            if n_kxc:
                b_kxc = 8.0 * N * min(chunk, Cn)
                roof_asm = {"bound": "hbm", "kernel": "k_kernel_matrix<.,false,.> K(X,C) chunk", "unit": "GB/s", "peak": 8000.0,
                            "achieved": b_kxc / (t_kxc * 1e-3 / n_kxc) / 1e9, "bytes_per_launch": b_kxc,
                            "avg_launch_ms": t_kxc / n_kxc, "launches": n_kxc}
                roof_asm["frac"] = roof_asm["achieved"] / roof_asm["peak"]
            if n_kxx:
                n_mat = len(thetas) + 1                                       # per cycle: the fit's evaluations + the refactor
                b_kxx = 8.0 * 128 * 128 * nb_ * (nb_ + 1) / 2
                roof_asm = dict(roof_asm or {}, kxx={"kernel": "k_kernel_matrix<.,true,.> K(X,X) lower tiles",
                                                     "achieved": n_mat * b_kxx / (t_kxx * 1e-3) / 1e9, "unit": "GB/s",
                                                     "bytes_per_matrix": b_kxx, "matrices": n_mat, "launches": n_kxx,
                                                     "total_ms": t_kxx,
                                                     "frac": n_mat * b_kxx / (t_kxx * 1e-3) / 1e9 / 8000.0})
            if roof_asm:
                roof_asm["note"] = ("HIP events on the handle's stream (BOBE_PROF_KXC / _KXX) over one cycle; fp64-VALU-bound "
                                    "in practice (~55 fp64 instructions per element incl. exp), DESIGN.md 4; PMC traffic: "
                                    "profiles/r05_traffic_k_kernel_matrix_*.json")
            t_cv, n_cv = class_ms["crossvv"]
            if n_cv:
                f_cv = 2.0 * N * M * min(chunk, Cn)
                roof_cross = {"bound": "mfma", "kernel": "k_cross_vv", "achieved": f_cv / (t_cv * 1e-3 / n_cv) / 1e12,
                              "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "avg_launch_ms": t_cv / n_cv, "launches": n_cv,
                              "flops_per_launch": f_cv}
                roof_cross["frac"] = roof_cross["achieved"] / FP64_MFMA_PEAK_TFLOPS
        chol, potrf_ms, roof_fit = {}, None, None
        flops_potrf = N ** 3 / 3.0
        if secondary:
            # Cholesky: mean device time of the factorisation alone (HIP events on the handle's stream): a lone one, B
            # advancing in lock step through one batched launch sequence, and (round 1's form) B on private streams
            ms_ = C.c_double()
            _lib.check(lib.bobe_debug_time_potrf(h, 10, C.byref(ms_)), "time_potrf")
            potrf_ms = ms_.value
            for B in (4, 8):
                _lib.check(lib.bobe_debug_time_potrf_lockstep(h, B, 10, C.byref(ms_)), "time_potrf_lockstep")
                chol[f"lockstep_{B}"] = {"in_flight": B, "ms_all": ms_.value, "gflops": B * flops_potrf / (ms_.value * 1e-3) / 1e9,
                                         "frac_of_fp64_mfma_peak": B * flops_potrf / (ms_.value * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
            _lib.check(lib.bobe_debug_time_potrf_batch(h, 4, 10, C.byref(ms_)), "time_potrf_batch")
            chol["streams_4"] = {"in_flight": 4, "ms_all": ms_.value, "gflops": 4 * flops_potrf / (ms_.value * 1e-3) / 1e9,
                                 "frac_of_fp64_mfma_peak": 4 * flops_potrf / (ms_.value * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
            # the fit phase as a whole: 20 value+gradient evaluations = 20 N^3 flops (potrf N^3/3 + inverse N^3/3 + K^-1/gradient N^3/3)
            fit_s = sub_ms["fit"] * 1e-3
            fit_ach = len(thetas) * float(N) ** 3 / fit_s / 1e12
            roof_fit = {"bound": "mfma", "phase": f"fit: {len(thetas)} x value+gradient ({key})", "achieved": fit_ach,
                        "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fit_ach / FP64_MFMA_PEAK_TFLOPS,
                        "flops": len(thetas) * float(N) ** 3, "ms": sub_ms["fit"],
                        "share_of_cycle": sub_ms["fit"] / (sub_ms["fit"] + sub_ms["refactor"] + sub_ms["sweep"]),
                        "potrf_ms_alone": potrf_ms, "potrf_frac_of_peak_alone": flops_potrf / (potrf_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
        cfg_name = {"headline": "(BASELINE.json configs[2])", "small": "(BASELINE.json configs[1])",
                    "shard": "(BASELINE.json configs[3])"}.get(args.config, "(not a BASELINE.json config)")
        out = {
            "metric": f"GP fit+acquisition cycles/sec at N={N} d={d}, {c_total if strong else Cn} cands; Cholesky GF/s",
            "value": (1 if strong else world) * args.steps / elapsed, "unit": "cycles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic RBF GP N={N} d={d}, "
                                   + (f"{c_total} candidates split over {world} GPU(s)" if strong else f"{Cn} candidates per GPU")
                                   + f", M={M}, fp64 {cfg_name}",
                       "N": N, "d": d, "candidates_per_gpu": Cn, "candidates_total": c_total, "M": M,
                       "evals_per_cycle": len(thetas),
                       "fit": (f"{R_total} restarts x {len(thetas) // R_total} value+gradient evaluations, restarts concurrent ({args.fit_mode})"
                               + (f", split over the ranks (this rank: restarts {my_restarts})" if strong and world > 1 else ""))
                       if R_total > 1 else f"{len(thetas)} sequential value+gradient evaluations",
                       "parallelism": f"candidate-sharded x{world}" + (", restart-sharded fit" if strong else ""),
                       "backend": args.backend if world > 1 else None, "exchange": args.exchange},
            "cholesky_gflops": flops_potrf / (potrf_ms * 1e-3) / 1e9 if potrf_ms else None,
            "cholesky_ms": potrf_ms,
            "cholesky_concurrent": chol.get("lockstep_4"),
            "cholesky": chol,
            "fit_ms": fit_ms,
            "sub_ms": sub_ms,
            "lbfgs_fit": lbfgs,
            "check": last if shard_rec is None else weak_check,
            "roofline": roof,
            "roofline_fit": roof_fit,
            "roofline_assembly": roof_asm,
            "roofline_cross": roof_cross,
            "accurate_sweep": accurate,
            "reference_noise_cycle": ref_noise,
            "cycles_per_s_matern": matern["cycles_per_s"] if matern else None,
            "matern_cycle": matern,
            # what rank 0 saw of the job: ranks, who carried the collectives
            "world_size": (dist.get_world_size() if world > 1 else 1),
            "backend": (str(dist.get_backend()) if world > 1 else None),
        }
        if shard_rec is not None:
            out["shard"] = shard_rec
        if not args.no_cpu_baseline and world == 1 and secondary:
            out["cpu_baseline"] = cpu_baseline(X, y, cand, Z, thetas, noise, gpu_check)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_rccl:
        mgpu.finalize()
    if world > 1:
        dist.barrier()              # rank 0's secondary measurements are done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
