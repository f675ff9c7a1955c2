"""The reference's one held number for the acquisition half - 'Mean acquisition value 3.4146e+00 at new points', iteration 1 of
its notebook run - against the GPU path's draws of the same quantity (tests/test_gpu_reference_held.py).  Prints the table
committed as profiles/r06_first_acquisition_value.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_reference_held as G  # noqa: E402
import test_reference_held_cpu as T  # noqa: E402
from bobe_amd.acquisition import WIPStd, get_mc_samples  # noqa: E402
from oracle import bobe_oracle as O  # noqa: E402
from oracle import bobe_oracle_loop as OL  # noqa: E402

held = G.NB["logged_mean_acquisition_values"]["iteration_1_to_15"][0]
b = G._notebook_bobe(G.NB["seed"])
gp = b.gp
og = O.OracleGP(gp.train_x, gp.train_y * gp.y_std + gp.y_mean, lengthscales=gp.lengthscales, kernel_variance=gp.kernel_variance)
acq = WIPStd()


def gpu_batch(g, mc, r):
    return acq.get_next_batch(g, n_batch=2, acq_kwargs={"mc_samples": {"x": mc}, "mc_points_size": 64}, n_restarts=1, maxiter=100,
                              early_stop_patience=10, verbose=False, rng=r)[1]


got = T.first_acquisition_values(gp, gpu_batch)
ref = T.first_acquisition_values(og, lambda g, mc, r: OL.get_next_batch(g, "wipstd", mc, 64, 2, r)[1])
own = []
for s in range(8):
    r = np.random.default_rng(100 + s)
    mc = get_mc_samples(gp, warmup_steps=512, num_samples=512, thinning=4, method="NUTS", num_chains=4, np_rng=r)
    own.append(float(np.mean(gpu_batch(gp, mc["x"], r))))
own = np.array(own)
print(f"# iteration 1 of examples/Example Notebook.ipynb (seed {G.NB['seed']}): y_std {gp.y_std:.2f}, hyper-parameters {gp.hyperparams_dict()}")
print(f"# the reference logged: Mean acquisition value {held:.4e} at new points (WIPStd, batch of 2, 64 integration points)")
print("seed   GPU path (exact posterior samples)   oracle (same samples)   relative difference")
for s, a, c in zip(T.FIRST_ACQ_SEEDS, got, ref):
    print(f"{s:4d}   {a:12.6f}                          {c:12.6f}            {abs(a - c) / c:.1e}")
print(f"GPU path: mean {got.mean():.3f}, standard deviation {got.std():.3f}, range {got.min():.3f} ... {got.max():.3f}; the logged value sits "
      f"{(held - got.mean()) / got.std():+.2f} sigma from the mean")
print(f"with the product's own sampler (HMC chains on the device, run()'s defaults), eight seeds: {np.round(own, 3).tolist()}, mean {own.mean():.3f}")
