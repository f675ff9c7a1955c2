"""ORACLE REGRESSION, not reference parity: the committed .npz vectors were written by the oracle itself
(tests/golden/make_golden.py), so this file only guards the oracle against drifting.  What pins the oracle to the
reference is tests/test_reference_held_cpu.py (the reference's own logged fit) and the independent implementations of
tests/test_oracle.py.  Should somebody with the reference's JAX environment run tests/golden/make_reference_golden.py and
commit its ref_*.npz, those files are picked up here too (tolerances allow for XLA-vs-LAPACK rounding)."""
import glob
import os

import numpy as np
import pytest

from oracle import bobe_oracle as O

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("loop_"))     # loop_*: BO-step fixtures (tests/test_gpu_loop_parity.py)


def load(path):
    g = dict(np.load(path, allow_pickle=False))
    g["prior"] = None if str(g["prior"]) == "none" else str(g["prior"])
    g["kernel"] = str(g["kernel"])
    return g


def oracle_gp(g):
    return O.OracleGP(g["X"], g["y"], noise=float(g["noise"]), kernel=g["kernel"], lengthscales=g["lengthscales"],
                      kernel_variance=float(g["kernel_variance"]), lengthscale_prior=g["prior"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    g = load(path)
    gp = oracle_gp(g)
    f, gr = gp.neg_mll_value_and_grad(g["theta"])
    assert f == pytest.approx(float(g["neg_mll"]), rel=1e-11)
    assert np.allclose(gr, g["neg_mll_grad"], rtol=1e-7, atol=1e-7 * np.max(np.abs(g["neg_mll_grad"])))
    assert np.allclose(gp.cholesky, g["cholesky"], rtol=0, atol=1e-10)
    sw = O.wip_sweep(gp, g["cand"], g["Z"])
    assert np.allclose(sw["wipv"], g["wipv"], rtol=1e-9, atol=1e-15)
    assert np.allclose(sw["wipstd"], g["wipstd"], rtol=1e-9, atol=1e-15)
    assert sw["argmin_v"] == int(g["argmin_v"]) and sw["argmin_s"] == int(g["argmin_s"])
    assert len(GOLDEN) >= 3
