"""Handle lifetime: everything a handle allocates on the device (factor buffers, evaluation slots, the lock-step batch's
workspace, launch plans, graphs, the classifier gate, sampler scratch) goes back when the handle is destroyed
(bobe_gp_destroy -> bobe_gp::release_all keeps a hand-written list of buffers: this test is what notices a missing entry).
The reference leaves this to Python's garbage collector (a GP is a bag of jax arrays, gp.py:248-281)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exercise(seed, n, d):
    """One handle through every entry point that owns device memory, then dropped."""
    from bobe_amd import GP, samplers
    from bobe_amd.clf_gp import GPwithClassifier
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = -30.0 * np.sum((X - 0.5) ** 2, axis=1)
    gp = GP(X, y, noise=1e-6, lengthscales=np.full(d, 0.5), kernel_variance=1.5)
    ls = np.full(d, 0.45)
    gp.mll_data(ls, 1.2)
    gp.mll_data(ls, 1.2, slot=1)                                                 # an evaluation slot (stream + graph)
    gp.mll_data_batch(np.tile(ls, (4, 1)) + 0.01 * np.arange(4)[:, None], np.full(4, 1.2))   # the lock-step workspace
    gp.fit(maxiter=3)
    cand, Z = rng.uniform(size=(1500, d)), rng.uniform(size=(200, d))
    gp.wip_sweep(cand, Z)
    gp.wip_grad(cand[:1], Z)
    gp.wip_grad(cand[:40], Z)
    gp.predict_batched(cand)
    gp.predict_grad(cand[:64])
    gp.acq_ei(cand[:64], float(np.max(gp.train_y)))
    gp.update(rng.uniform(size=(3, d)), -30.0 * rng.uniform(size=(3, 1)))        # rank-b append
    twin = gp.copy()                                                             # bobe_gp_clone_state
    twin.predict_mean_batched(cand[:16])
    samplers.sample_GP_NUTS(gp, np_rng=rng, warmup_steps=16, num_samples=32, thinning=4, num_chains=1)
    x0 = rng.uniform(0.05, 0.95, size=(64, d))
    gp.rwalk(x0, gp.predict_mean_batched(x0), 0.05 * np.eye(d), -1e30, 4, seed=seed)
    g = GPwithClassifier(X, y, clf_threshold=8.0, gp_threshold=20.0, noise=1e-6, lengthscales=np.full(d, 0.5),
                         minus_inf=-1e10)
    g.predict_mean_batched(cand[:64])
    del gp, twin, g
    gc.collect()


def test_destroyed_handles_return_their_device_memory():
    import torch
    _exercise(0, 600, 5)                                   # (first use: HIP context, code objects, the library's statics)
    _exercise(1, 1400, 5)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for s in range(2, 10):
        _exercise(s, 600 if s % 2 else 1400, 5)            # (two sizes: every buffer is re-sized on the way)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    # one leaked N x N buffer per handle would be 8 x (2.9 ... 15.7 MB); the allowance is the allocator's granularity
    assert free0 - free1 < 4 << 20, f"device memory not returned: {(free0 - free1) / 2 ** 20:.1f} MiB over eight handles"
    # the measurement sees what a handle holds: one kept alive shows up, and goes away with it
    from bobe_amd import GP
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(2000, 5))
    kept = GP(X, np.sin(X.sum(1)), noise=1e-6, lengthscales=np.full(5, 0.5))
    kept.mll_data(np.full(5, 0.5), 1.0)
    torch.cuda.synchronize()
    held = free1 - torch.cuda.mem_get_info(0)[0]
    assert held > 3 * 2000 * 2000 * 8, f"a live N = 2000 handle shows as {held / 2 ** 20:.1f} MiB only"
    del kept
    gc.collect()
    torch.cuda.synchronize()
    assert abs(free1 - torch.cuda.mem_get_info(0)[0]) < 4 << 20
