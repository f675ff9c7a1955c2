"""Acquisition functions on top of the GPU GP — counterpart of BOBE/acquisition.py.

WIPV / WIPStd evaluate every candidate in ONE call of ``bobe_gp_wip_sweep`` (the reference maps
``fun`` over the candidates sequentially with ``lax.map``, acquisition.py:390-394).  EI / LogEI are
pointwise scorers on the batched posterior (``bobe_gp_acq_ei``).

Where the reference differentiates ``fun`` with JAX the build has explicit gradient entry points: EI / LogEI
restarts (acquisition.py:281-290) use the analytic posterior gradients of ``bobe_gp_predict_grad``, the WIPV /
WIPStd local refinement for N <= 500 (acquisition.py:403-412) the analytic score gradients of
``bobe_gp_wip_grad``.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np
from scipy.stats import qmc

from . import _lib
from .gp import GP
from .optim import optimize_optax, optimize_scipy
from .utils import get_logger, get_numpy_rng

log = get_logger("acq")



class AcquisitionFunction:
    """BOBE/acquisition.py:79-196."""

    name: str = "BaseAcquisitionFunction"

    def __init__(self, optimizer: str = "scipy", optimizer_options: Optional[Dict[str, Any]] = None):
        self.optimizer = optimizer
        self.optimizer_options = optimizer_options if optimizer_options is not None else {}
        self.acq_optimize = optimize_scipy if optimizer == "scipy" else optimize_optax      # acquisition.py:101-104

    def fun(self, x, *args, **kwargs):
        raise NotImplementedError

    def get_next_point(self, gp: GP, acq_kwargs=None, maxiter=500, n_restarts=8, verbose=True,
                       early_stop_patience=25, rng=None):
        raise NotImplementedError("Base class get_next() not implemented")

    def get_next_batch(self, gp: GP, n_batch: int = 1, acq_kwargs=None, maxiter: int = 500, n_restarts: int = 8,
                       verbose: bool = True, early_stop_patience: int = 25, rng=None):
        """Kriging-believer batch (BOBE/acquisition.py:147-196)."""
        rng = rng if rng is not None else get_numpy_rng()
        acq_kwargs = acq_kwargs if acq_kwargs is not None else {}
        x_batch, acq_vals = [], []
        x_next, val = self.get_next_point(gp, acq_kwargs=acq_kwargs, maxiter=maxiter, n_restarts=n_restarts,
                                          verbose=verbose, early_stop_patience=early_stop_patience, rng=rng)
        x_batch.append(np.asarray(x_next))
        acq_vals.append(val)
        if n_batch > 1:
            dummy_gp = _believer_gp(gp)                                        # acquisition.py:175-180
            dummy_gp.update(x_next, dummy_gp.predict_mean_single(x_next))
            for member in range(1, n_batch):
                x_next, val = self.get_next_point(dummy_gp, acq_kwargs=acq_kwargs, maxiter=maxiter,
                                                  n_restarts=n_restarts, verbose=verbose,
                                                  early_stop_patience=early_stop_patience, rng=rng)
                x_batch.append(np.asarray(x_next))
                acq_vals.append(val)
                if member + 1 < n_batch:      # (the reference also updates after the LAST member, acquisition.py:194: the dummy
                    dummy_gp.update(x_next, dummy_gp.predict_mean_single(x_next))     # GP is dropped right after - not done here)
        return np.array(x_batch), np.array(acq_vals)


def _believer_gp(gp: GP) -> GP:
    """The plain GP the kriging believer works on (acquisition.py:175-180: same data, kernel and hyper-parameters as ``gp``,
    default priors).  The reference constructs a new one per batch, i.e. standardises and factorises again; here ONE scratch
    GP per surrogate is kept and brought to ``gp``'s state on the device (``bobe_gp_clone_state``: data, L, L^-1, alpha
    copied, no factorisation).  Creating and destroying a handle costs ~40 hipMalloc / hipFree calls - 6 of the 16 ms of a
    batch of five at N = 400 (tools/acq_host_profile.py)."""
    if gp._pushed_hyper != gp._hyper_key():          # attributes changed by hand since the last factorisation: the reference's
        return GP(train_x=gp.train_x, train_y=gp.train_y * gp.y_std + gp.y_mean, noise=gp.noise, kernel=gp.kernel_name,
                  lengthscales=gp.lengthscales, kernel_variance=gp.kernel_variance, device=gp.device)   # dummy sees the new ones
    d = getattr(gp, "_believer_scratch", None)
    if d is None or d.ndim != gp.ndim or d.kernel_name != gp.kernel_name or d.device != gp.device:
        d = GP(train_x=gp.train_x, train_y=gp.train_y * gp.y_std + gp.y_mean, noise=gp.noise, kernel=gp.kernel_name,
               lengthscales=gp.lengthscales, kernel_variance=gp.kernel_variance, device=gp.device, _factor=False)
        gp._believer_scratch = d
    d.train_x, d.train_y = np.array(gp.train_x), np.array(gp.train_y)
    d.y_mean, d.y_std = gp.y_mean, gp.y_std
    d.lengthscales, d.kernel_variance, d.noise = np.array(gp.lengthscales), float(gp.kernel_variance), float(gp.noise)
    _lib.check(d._lib.bobe_gp_clone_state(d._h, gp._h), "bobe_gp_clone_state")
    d.not_pd = bool(gp.not_pd)
    d._pushed_hyper = gp._pushed_hyper
    d._chol_cache = d._alpha_cache = None
    return d


class EI(AcquisitionFunction):
    """BOBE/acquisition.py:199-291."""

    name: str = "EI"
    _log = False

    def fun(self, x, gp, best_y, zeta):
        """-EI(x) (acquisition.py:226-253); x may be (d,) or (C, d)."""
        val = -gp.acq_ei(np.atleast_2d(x), best_y, zeta, log_ei=self._log)
        return val[0] if np.ndim(x) == 1 else val

    def _value_and_grad(self, gp, best_y, zeta):
        """(-EI, d(-EI)/dx) or (-logEI, ...) from the GPU's analytic posterior gradients (bobe_gp_predict_grad):
        dEI = Phi(u) dmu + phi(u) dsigma, u = (mu - zeta - best)/sigma; logEI through log-space ratios."""
        from scipy.special import log_ndtr
        lo = 1e-18 if self._log else 1e-20                      # acquisition.py:247, 324

        def vg(x):
            x = np.asarray(x, dtype=np.float64)
            m, v, dm, dv = gp.predict_grad(x[None, :])
            m, v, dm, dv = float(m[0]), float(v[0]), dm[0], dv[0]
            clipped = v < lo
            v = max(v, lo)
            sigma = np.sqrt(v)
            dsig = np.zeros_like(dv) if clipped else dv / (2.0 * sigma)
            u = (m - zeta - best_y) / sigma
            log_phi = -0.5 * u * u - 0.5 * np.log(2.0 * np.pi)
            log_Phi = float(log_ndtr(u))
            if not self._log:
                val = -float(gp.acq_ei(x[None, :], best_y, zeta, log_ei=False)[0])
                return val, -(np.exp(log_Phi) * dm + np.exp(log_phi) * dsig)
            log_ei = float(gp.acq_ei(x[None, :], best_y, zeta, log_ei=True)[0])       # = log h(u) + log sigma
            # d logEI = (Phi dmu + phi dsigma) / EI, with EI = exp(log_ei)
            grad = np.exp(log_Phi - log_ei) * dm + np.exp(log_phi - log_ei) * dsig
            return -log_ei, -grad
        return vg

    def get_next_point(self, gp, acq_kwargs=None, maxiter: int = 250, n_restarts: int = 20, verbose: bool = True,
                       early_stop_patience: int = 25, rng=None):
        rng = rng if rng is not None else get_numpy_rng()
        acq_kwargs = acq_kwargs if acq_kwargs is not None else {}
        zeta = acq_kwargs.get("zeta", 0.0)
        best_y = acq_kwargs.get("best_y", float(np.max(gp.train_y)))
        best_x = gp.train_x[int(np.argmax(gp.train_y))]
        if n_restarts > 1:                                                     # acquisition.py:271-278
            n_random = int(n_restarts / 2)
            x0 = np.vstack([gp.get_random_point(rng, nstd=5) for _ in range(n_random)])
            x0 = np.vstack([x0, np.full((n_restarts - n_random, gp.ndim), best_x)])
        else:
            x0 = np.atleast_2d(best_x)
        x0 = np.clip(x0 + rng.normal(0.0, 0.005, size=x0.shape), 0.0, 1.0)
        pts, vals = self.acq_optimize(self._value_and_grad(gp, best_y, zeta), num_params=gp.ndim, x0=x0,
                                      bounds=[0, 1], optimizer_options=dict(self.optimizer_options),
                                      maxiter=maxiter, n_restarts=n_restarts, verbose=verbose)
        return pts, -vals


class LogEI(EI):
    """BOBE/acquisition.py:293-330."""

    name: str = "LogEI"
    _log = True


class WeightedIntegratedPosteriorBase(AcquisitionFunction):
    """BOBE/acquisition.py:333-412."""

    _key = "wipv"

    def fun(self, x, gp, mc_points=None, k_train_mc=None):
        r = gp.wip_sweep(np.atleast_2d(x), mc_points)
        return r[self._key][0] if np.ndim(x) == 1 else r[self._key]

    def sweep(self, gp, candidates, mc_points):
        """Scores of all candidates and the argmin (acquisition.py:394-398)."""
        r = gp.wip_sweep(candidates, mc_points)
        idx = r["argmin_v"] if self._key == "wipv" else r["argmin_s"]
        return r[self._key], idx

    def sweep_best(self, gp, candidates, mc_points, group=None):
        """(best score, index of the best candidate).  With an initialised ``torch.distributed`` group of G > 1
        ranks every rank scores its contiguous shard of the candidates on its own GPU (same factor everywhere) and
        one all-gather of (min score, global index) picks the winner, ties to the lowest index like
        ``jnp.argmin`` (acquisition.py:397)."""
        from .dist_sweep import dist_info, sharded_wip_sweep
        world, _, coll_dev = dist_info(group, gp.device)
        if world > 1 and candidates.shape[0] >= world:
            _, gmin, gidx = sharded_wip_sweep(lambda c: self.sweep(gp, c, mc_points), candidates, group=group,
                                              device=coll_dev)
            return float(gmin), int(gidx)
        vals, idx = self.sweep(gp, candidates, mc_points)
        return float(vals[idx]), int(idx)

    def get_next_point(self, gp, acq_kwargs=None, maxiter: int = 100, n_restarts: int = 1, verbose: bool = True,
                       early_stop_patience: int = 25, rng=None):
        acq_kwargs = acq_kwargs if acq_kwargs is not None else {}
        mc_samples = acq_kwargs.get("mc_samples")
        mc_points_size = acq_kwargs.get("mc_points_size", 128)
        mc_points = get_mc_points(mc_samples, mc_points_size=mc_points_size, rng=rng)
        best_val, idx = self.sweep_best(gp, mc_points, mc_points)            # candidates == integration points
        best_x = np.array(mc_points[idx])
        if gp.train_x.shape[0] > 500:                                          # acquisition.py:400-401
            return best_x, best_val

        def vg(x):                                    # value and exact gradient in one call (bobe_gp_wip_grad)
            wv, ws, dv, ds = gp.wip_grad(np.asarray(x, dtype=np.float64)[None, :], mc_points)
            return (float(wv[0]), dv[0]) if self._key == "wipv" else (float(ws[0]), ds[0])
        return self.acq_optimize(vg, num_params=gp.ndim, x0=best_x, bounds=[0, 1],
                                 optimizer_options=dict(self.optimizer_options), maxiter=maxiter,
                                 n_restarts=n_restarts, verbose=verbose)


class WIPV(WeightedIntegratedPosteriorBase):
    """BOBE/acquisition.py:415-440."""
    name: str = "WIPV"
    _key = "wipv"


class WIPStd(WeightedIntegratedPosteriorBase):
    """BOBE/acquisition.py:443-465."""
    name: str = "WIPStd"
    _key = "wipstd"


def get_mc_samples(gp: GP, warmup_steps=512, num_samples=1024, thinning=4, method="NUTS", num_chains=4,
                   np_rng=None, rng_key=None):
    """BOBE/acquisition.py:468-482: 'NUTS' (Hamiltonian Monte Carlo on the surrogate, batched on the GPU), 'NS'
    (nested sampling on the surrogate) or 'uniform' (scrambled Sobol)."""
    if method == "uniform":
        return {"x": qmc.Sobol(gp.ndim, scramble=True, seed=np_rng).random(num_samples)}
    if method == "NS":                                   # acquisition.py:473-475, batched on the GPU GP
        from .samplers import nested_sampling
        rng = np_rng if isinstance(np_rng, np.random.Generator) else np.random.default_rng(np_rng)
        samples, _, _ = nested_sampling(gp, ndim=gp.ndim, mode="acq", rng=rng)
        return samples
    if method == "NUTS":                                 # acquisition.py:470-472, batched HMC on the GPU GP
        from .samplers import sample_GP_NUTS
        rng = np_rng if isinstance(np_rng, np.random.Generator) else np.random.default_rng(np_rng)
        return sample_GP_NUTS(gp, np_rng=rng, rng_key=rng_key, num_chains=num_chains, warmup_steps=warmup_steps,
                              num_samples=num_samples, thinning=thinning)
    raise ValueError(f"Unknown method {method} for sampling GP")          # acquisition.py:481


def get_mc_points(mc_samples, mc_points_size=128, rng=None):
    """BOBE/acquisition.py:485-489."""
    mc_size = max(mc_samples["x"].shape[0], mc_points_size)
    rng = rng if rng is not None else get_numpy_rng()
    idxs = rng.choice(mc_size, size=mc_points_size, replace=False)
    return mc_samples["x"][idxs]
