"""One rank of the two-process RCCL rehearsal of tests/test_gpu_dist.py: both ranks on the box's ONE GPU, the library's own
communicator (bobe_mgpu_init) carrying two ranks.  A gloo group ships the unique id and the verdicts; rank 0 writes what
happened as JSON to argv[1].  If RCCL refuses two ranks on one device the refusal text is the result."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bobe_amd import GP, mgpu  # noqa: E402
from bobe_amd.dist_sweep import shard_bounds  # noqa: E402


def problem():
    rng = np.random.default_rng(5)
    n, d = 300, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, 2]
    cand, Z = rng.uniform(size=(2001, d)), rng.uniform(size=(64, d))
    return X, y, cand, Z


if __name__ == "__main__":
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    X, y, cand, Z = problem()
    gp = GP(X, y, noise=1e-6, lengthscales=np.full(3, 0.5), kernel_variance=1.2)
    out = {"world": world, "ok": False, "error": None}
    try:
        mgpu.init_from_torch(device=0)
        out["ok"] = mgpu.world() == world
    except Exception as e:                                   # RCCL's own refusal (e.g. two ranks on one device)
        out["error"] = str(e)
    flags = [None] * world
    dist.all_gather_object(flags, out["ok"])
    if all(flags):
        lo, hi = shard_bounds(cand.shape[0], world, rank)
        r = mgpu.wip_sweep(gp, cand[lo:hi], lo, Z)
        bm, bt = mgpu.best_fit(-10.0 - rank, np.array([0.1 * (rank + 1), 0.2, 0.3]))     # rank 0 holds the better fit
        out.update(argmin_v=int(r["argmin_v"]), argmin_s=int(r["argmin_s"]), min_v=float(r["min_v"]), min_s=float(r["min_s"]),
                   best_mll=float(bm), best_theta=np.asarray(bt).tolist(), shard=[lo, hi])
        mgpu.finalize()
    else:
        try:
            mgpu.finalize()
        except Exception:
            pass
    every = [None] * world
    dist.all_gather_object(every, out)
    if rank == 0:
        with open(sys.argv[1], "w") as fh:
            json.dump(every, fh)
    dist.barrier()
    dist.destroy_process_group()
