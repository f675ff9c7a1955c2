"""world_size-2 gloo test of the candidate-sharded sweep exchange (bobe_amd/dist_sweep.py).
The per-shard scorer here is the CPU oracle (test infrastructure) — the exchange / tie-break /
shard arithmetic under test is the product code that runs over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bobe_amd.dist_sweep import merge_argmin, merge_best_fit, shard_bounds, sharded_wip_sweep


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            ref = np.array_split(np.arange(n), w)
            assert [hi - lo for lo, hi in spans] == [len(x) for x in ref]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import bobe_oracle as O
        rng = np.random.default_rng(11)
        n, d, c, m = 80, 3, 101, 24
        X = rng.uniform(size=(n, d))
        y = np.sin(4 * X[:, 0]) - X[:, 1] * X[:, 2]
        gp = O.OracleGP(X, y, noise=1e-6, lengthscales=[0.3, 0.4, 0.5])
        cand, Z = rng.uniform(size=(c, d)), rng.uniform(size=(m, d))
        cand[70] = cand[5]          # duplicate candidate -> tie across shards, lowest index must win

        def score(shard):
            r = O.wip_sweep(gp, shard, Z)
            return r["wipstd"], r["argmin_s"]
        scores, gmin, gidx = sharded_wip_sweep(score, cand, device=None)
        full = O.wip_sweep(gp, cand, Z)
        ok = (gidx == full["argmin_s"]) and abs(gmin - full["wipstd"].min()) < 1e-15
        # forced tie
        tmin, tidx = merge_argmin(0.5, 10 if rank == 1 else 40)
        ok = ok and (tmin == 0.5 and tidx == 10)
        # NaN beats numbers (argmin propagates NaN)
        nmin, nidx = merge_argmin(float("nan") if rank == 1 else 0.1, 7 + rank)
        ok = ok and np.isnan(nmin) and nidx == 8
        mll, par = merge_best_fit(1.0 + rank if rank == 0 else float("nan"), np.array([rank, 2.0 * rank]))
        ok = ok and mll == 1.0 and np.array_equal(par, [0.0, 0.0])
        mll, par = merge_best_fit(-3.0 + 5 * rank, np.array([rank, 2.0 * rank]))
        top = world - 1                                  # the largest mll sits on the last rank
        ok = ok and mll == -3.0 + 5 * top and np.array_equal(par, [top, 2.0 * top])
        # the 8-GPU shape of the strong mode (bench.py --gpus 8): the fit has 4 restarts, so ranks 4.. hold none and hand in
        # (-inf, theta_0) - it must never win, not even against a restart whose mll is NaN or very negative
        from bobe_amd.dist_sweep import shard_bounds as sb
        lo, hi = sb(4, world, rank)
        mine = (-1e300 if rank == 0 else -50.0 - rank, np.array([float(rank), 1.0])) if hi > lo else \
            (-np.inf, np.array([99.0, 99.0]))
        mll, par = merge_best_fit(*mine)
        holders = [r for r in range(world) if sb(4, world, r)[1] > sb(4, world, r)[0]]
        best = max(holders, key=lambda r: (-1e300 if r == 0 else -50.0 - r))
        ok = ok and par[0] == float(best) and par[0] != 99.0 and np.isfinite(mll)
        # ... and a rank without candidates (5 candidates over 8 ranks) hands in (+inf, 2^52): never the argmin
        lo, hi = sb(5, world, rank)
        loc = (10.0 + rank, lo) if hi > lo else (float("inf"), 2 ** 52)
        smin, sidx = merge_argmin(*loc)
        ok = ok and smin == 10.0 and sidx == 0
        q.put((rank, bool(ok), int(gidx), int(full["argmin_s"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_sweep_gloo(world):
    """two ranks, three (uneven shards of the 101 candidates: 34 / 34 / 33, np.array_split's bounds) and eight - the shape
    of the driver's 8-GPU run: more ranks than restarts, ranks without a restart or without a candidate in the merges"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, gidx, want in res:
        assert ok, (rank, gidx, want)
