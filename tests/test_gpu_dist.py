"""The N > 1 product path on real kernels: a 2-rank (gloo) BO run — restarts split over the ranks
(bobe_amd.bo.gp_fit, pool.py:298-326) and candidate-sharded WIPV sweeps with the (min, index) all-gather
(acquisition.sweep_best) — must retrace the single-process run.  Both ranks share the box's one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bo_run_retraces_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    import dist_bo_worker as W
    single = W.run_case()
    out = tmp_path / "dist.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:          # a free port: two suites may share a box
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "workers", "dist_bo_worker.py"), str(out), str(tmp_path / "run")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    with open(out) as fh:
        two = json.load(fh)
    assert two["n"] == single["n"]
    # same acquisition choices (the shard merge keeps argmin semantics) and the same fitted hyper-parameters
    # (the union of the ranks' restarts is the single-process restart set; max-by-mll picks the same optimum)
    assert np.allclose(two["train_x"], single["train_x"], atol=1e-12)
    assert np.allclose(two["lengthscales"], single["lengthscales"], rtol=1e-9)
    assert two["kernel_variance"] == pytest.approx(single["kernel_variance"], rel=1e-9)
    assert two["best_val"] == pytest.approx(single["best_val"], rel=1e-12)
    # the two ranks ran with the constructor's default save=True into ONE directory: rank 0 alone wrote, nothing half-written
    # is left, and the files resume (every rank used to write the same temporary names)
    files = sorted(os.listdir(tmp_path / "run"))
    assert "banana_gp.npz" in files and "banana_run.json" in files and not [f for f in files if "tmp" in f], files
    from bobe_amd import GP
    assert GP.load(str(tmp_path / "run" / "banana_gp")).npoints == two["n"]
    with open(tmp_path / "run" / "banana_run.json") as fh:
        st = json.load(fh)
    assert st["gp_file"] in files and st["gp_training_set_size"] == two["n"]


def test_library_owned_rccl_exchange_with_two_ranks_on_the_one_gpu(tmp_path):
    """bobe_mgpu_init / _wip_sweep / _best_fit with world = 2: two processes, both on the box's one GPU, the library's own RCCL
    communicator between them - so that an 8-GPU node is not RCCL's first multi-rank run of this code.  RCCL may refuse two
    ranks on one device ("Duplicate GPU detected"): then the refusal must reach BOTH ranks as an error code with RCCL's text
    (no hang, no crash), it is recorded in gpurun_out/r06_rccl_two_ranks_one_gpu.txt, and the test says so."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    import rccl_two_rank_worker as W
    out = tmp_path / "rccl.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "workers", "rccl_two_rank_worker.py"), str(out)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    with open(out) as fh:
        ranks = json.load(fh)
    assert len(ranks) == 2
    note = os.path.join(ROOT, "gpurun_out", "r06_rccl_two_ranks_one_gpu.txt")
    os.makedirs(os.path.dirname(note), exist_ok=True)
    if all(r["ok"] for r in ranks):
        from bobe_amd import GP
        X, y, cand, Z = W.problem()
        gp = GP(X, y, noise=1e-6, lengthscales=np.full(3, 0.5), kernel_variance=1.2)
        ref = gp.wip_sweep(cand, Z)
        for r in ranks:                                         # every rank holds the global result
            assert r["argmin_v"] == ref["argmin_v"] and r["argmin_s"] == ref["argmin_s"]
            assert r["min_s"] == ref["wipstd"][ref["argmin_s"]] and r["min_v"] == ref["wipv"][ref["argmin_v"]]
            assert r["best_mll"] == -10.0 and r["best_theta"] == [0.1, 0.2, 0.3]
        assert ranks[0]["shard"] == [0, 1001] and ranks[1]["shard"] == [1001, 2001]
        with open(note, "w") as fh:
            fh.write("two ranks on ONE MI355X through the library's own RCCL communicator: bobe_mgpu_init(world=2) accepted, "
                     "bobe_mgpu_wip_sweep and bobe_mgpu_best_fit returned the single-process result on both ranks\n")
    else:
        texts = [r["error"] for r in ranks]
        assert all(t for t in texts), ranks                     # every rank got an error code with a text, none hung
        with open(note, "w") as fh:
            fh.write("two ranks on ONE MI355X: RCCL refuses (bobe_mgpu_init error text per rank):\n" + "\n".join(texts) + "\n")
        assert all("nccl" in t.lower() or "rccl" in t.lower() for t in texts), texts
