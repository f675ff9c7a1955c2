"""Snapshot / compare the bits of factorisation-dependent outputs (before/after a kernel change that must not move them).
   python tools/bits_snapshot.py save|check FILE        (BITS_MAX_N=1500 skips the large sizes; `print` writes the digests
   as one JSON line to stdout)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd.gp import GP  # noqa: E402

out = {}
MAX_N = int(os.environ.get("BITS_MAX_N", 1 << 30))
SIZES = ((17, 2, "rbf"), (100, 3, "matern"), (129, 4, "rbf"), (641, 5, "rbf"), (1500, 8, "matern"), (2048, 8, "rbf"), (4096, 8, "rbf"))
if os.environ.get("BITS_SIZES"):          # e.g. BITS_SIZES=3000:5:matern,3500:8:rbf
    SIZES = tuple((int(a), int(b), c) for a, b, c in (t.split(":") for t in os.environ["BITS_SIZES"].split(",")))
for N, d, kern in SIZES:
    if N > MAX_N:
        continue
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
    gp = GP(X, y, noise=1e-5, kernel=kern, lengthscales=np.full(d, 0.5))
    out[f"L_{N}"] = np.array(gp.cholesky)
    m, g = gp.mll_data(np.full(d, 0.45), 1.3)
    out[f"m_{N}"], out[f"g_{N}"] = np.array(m), np.array(g)
    B = 4
    lsb = np.full((B, d), 0.4) + 0.03 * np.arange(B)[:, None]
    mb, gb = gp.mll_data_batch(lsb, np.ones(B))
    out[f"mb_{N}"], out[f"gb_{N}"] = np.array(mb), np.array(gb)
    if os.environ.get("BITS_B8"):            # an eight-wide batch (its first panels do not fit one launch): the first four
        ls8 = np.vstack([lsb, lsb + 0.2])    # members are the batch above and must return its bits
        m8, g8 = gp.mll_data_batch(ls8, np.ones(8))
        assert np.array_equal(m8[:4], mb) and np.array_equal(g8[:4], gb), "a batch member's bits depend on the batch width"
        out[f"m8_{N}"], out[f"g8_{N}"] = np.array(m8), np.array(g8)
    mu, var = gp.predict_batched(rng.uniform(size=(50, d)))
    out[f"mu_{N}"], out[f"var_{N}"] = np.array(mu), np.array(var)
    if N <= 2048:     # a sweep of three chunks (the assembly stream runs ahead of the GEMM launches from two chunks up)
        rs = np.random.default_rng(10_000 + N)
        gp._lib.bobe_gp_set_chunk(gp._h, 256)
        r = gp.wip_sweep(rs.uniform(size=(700, d)), rs.uniform(size=(96, d)), want_mean_var=True)
        gp._lib.bobe_gp_set_chunk(gp._h, 0)
        out[f"sw_{N}"] = np.concatenate([r["wipv"], r["wipstd"], r["mean"], r["var"], [r["argmin_v"], r["argmin_s"]]])
import hashlib
import json
out = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in out.items()}
if sys.argv[1] == "print":
    print(json.dumps(out))
elif sys.argv[1] == "save":
    json.dump(out, open(sys.argv[2], "w"), indent=0)
    print("saved", len(out), "digests")
else:
    ref = json.load(open(sys.argv[2]))
    bad = [k for k in out if k in ref and out[k] != ref[k]]
    print("not in the reference:", [k for k in out if k not in ref])
    print("BITS DIFFER in:" if bad else "all bits identical", bad)
    sys.exit(1 if bad else 0)
