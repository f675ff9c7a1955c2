"""One rank of the 2-rank BO run of tests/test_gpu_dist.py (launched with torch.distributed.run, gloo backend,
both ranks on the box's one GPU).  Rank 0 writes the trajectory as JSON to argv[1]; argv[2], when
given, is the run's save_dir (default save=True)."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bobe_amd.bo import BOBE  # noqa: E402


def banana(x):
    return -0.5 * (x[0] ** 2 / 4.0 + (x[1] - 0.25 * x[0] ** 2) ** 2 * 4.0)


def run_case(save_dir=None):
    """save_dir: run with the constructor's default ``save=True`` into that directory (every rank is handed the same one:
    only rank 0 may write there)."""
    bounds = np.array([[-4.0, 4.0], [-2.0, 6.0]]).T
    kw = dict(save=False) if save_dir is None else dict(save_dir=save_dir, likelihood_name="banana", save_step=1)
    bobe = BOBE(banana, ["x", "y"], bounds, n_sobol_init=12, seed=11, **kw)
    if save_dir is not None:
        assert bobe.is_main == (not dist.is_initialized() or dist.get_rank() == 0) and bobe.save == bobe.is_main
    res = bobe.run(acq="wipv", max_evals=22, mc_points_size=96, num_mc_samples=512, fit_n_points=2,
                   mc_points_method="uniform")
    gp = res["gp"]
    return {"train_x": gp.train_x.tolist(), "lengthscales": np.asarray(gp.lengthscales).tolist(),
            "kernel_variance": float(gp.kernel_variance), "best_val": res["best_val"], "n": int(gp.npoints)}


if __name__ == "__main__":
    dist.init_process_group("gloo")
    out = run_case(sys.argv[2] if len(sys.argv) > 2 else None)
    if dist.get_rank() == 0:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
