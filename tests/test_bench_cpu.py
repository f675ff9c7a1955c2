"""bench.py's launcher logic without a GPU: --gpus N builds the torchrun child before torch is imported, and a
mismatching torchrun environment is refused."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_starts_n_ranks_through_torchrun(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--config", "shard"])
    rc = bench.launch_ranks(bench.parse(["--gpus", "4", "--steps", "2", "--config", "shard"]))
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--config", "shard"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert 1024 < bench.free_port() < 65536


def test_world_size_mismatch_is_refused_before_any_gpu_work():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_cpu_port_of_the_baseline_matches_the_numpy_oracle():
    import numpy as np
    from oracle import bobe_oracle as O
    from oracle import cpu_port as P
    rng = np.random.default_rng(0)
    n, d = 300, 4
    X = rng.uniform(size=(n, d))
    y = np.sin(X.sum(1))
    y = (y - y.mean()) / y.std()
    ls = np.array([0.5, 0.6, 0.7, 0.4])
    m0, g0 = O.cycle_value_and_grad(X, y, ls, 1.3, 1e-6)
    m1, g1 = P.cycle_value_and_grad(X, y, ls, 1.3, 1e-6)
    assert abs(m0 - m1) <= 1e-11 * abs(m0) and np.max(np.abs(g0 - g1)) <= 1e-9 * np.max(np.abs(g0))
    og = O.OracleGP(X, y, noise=1e-6, lengthscales=ls, kernel_variance=1.3)
    cand, Z = rng.uniform(size=(200, d)), rng.uniform(size=(48, d))
    a = O.wip_sweep(og, cand, Z)
    b = P.wip_sweep(P.factor(X, og.train_y.reshape(-1), ls, 1.3, 1e-6), cand, Z, y_std=og.y_std)
    assert np.allclose(a["wipv"], b["wipv"], rtol=1e-8) and np.allclose(a["wipstd"], b["wipstd"], rtol=1e-8)
    assert np.allclose(a["mean"], b["mean"], atol=1e-8) and a["argmin_s"] == b["argmin_s"]
    assert P.host_description()["threads"] >= 1
