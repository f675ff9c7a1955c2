"""Candidate-sharded acquisition sweep across GPUs (one process per GPU, torch.distributed).

The reference has no candidate parallelism (sequential lax.map, BOBE/acquisition.py:394); the fit's
restart sharding it does have (BOBE/pool.py:298-326: array_split + max by mll) is mirrored by
``merge_best_fit``.  Each rank scores its contiguous shard [r*C/G, (r+1)*C/G) with its own GP handle
(identical factor on every rank: deterministic kernels, zero traffic), then ONE all-gather of
(min score, global index) decides the winner; ties go to the lowest global index, matching
jnp.argmin's first-occurrence rule (BOBE/acquisition.py:397).  Backend: "nccl" (= RCCL over xGMI)
on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import sys
from typing import Callable, Optional, Tuple

import numpy as np


def dist_info(group=None, gpu_index: Optional[int] = None):
    """(world size, rank, device for collective payloads) of the initialised process group, (1, 0, None) without
    one.  RCCL ("nccl") needs the payload on this rank's GPU; gloo takes host tensors."""
    if "torch.distributed" not in sys.modules:    # nobody can have initialised a process group: skip the (1 s) import
        return 1, 0, None
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = None
    if world > 1 and str(dist.get_backend(group)).lower() == "nccl":
        dev = torch.device("cuda", gpu_index if gpu_index is not None else torch.cuda.current_device())
    return world, rank, dev


def shard_bounds(n_candidates: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard of rank ``rank`` — np.array_split boundaries (BOBE/pool.py:302)."""
    base, extra = divmod(n_candidates, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_argmin(scores, global_idx) -> Tuple[float, int]:
    """The merge rule on the gathered per-shard (min score, global index) pairs: smallest score wins, ties go to the
    lowest global index (``jnp.argmin`` returns the first occurrence, acquisition.py:397); a NaN score counts as
    minimal (np.argmin / jnp.argmin propagate NaN)."""
    scores = np.asarray(scores, dtype=np.float64).reshape(-1)
    idx = np.asarray(global_idx, dtype=np.int64).reshape(-1)
    key = np.where(np.isnan(scores), -np.inf, scores)
    order = np.lexsort((idx, key))
    return float(scores[order[0]]), int(idx[order[0]])


def merge_argmin(local_min: float, local_global_idx: int, group=None, device=None) -> Tuple[float, int]:
    """ONE all-gather of (score, global index) per acquisition, then ``reduce_argmin`` on every rank.  The index
    travels in the same float64 payload as the score (16 B per rank, one collective): exact up to 2^53 candidates."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(local_min), int(local_global_idx)
    if not 0 <= int(local_global_idx) < 2 ** 53:
        raise ValueError("candidate index does not fit the float64 payload of the all-gather")
    world = dist.get_world_size(group)
    mine = torch.tensor([float(local_min), float(local_global_idx)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allv = torch.stack(parts).cpu().numpy()
    return reduce_argmin(allv[:, 0], allv[:, 1].astype(np.int64))


def merge_best_fit(local_mll: float, local_params: np.ndarray, group=None, device=None):
    """max-by-mll over ranks (BOBE/pool.py:322-326); non-finite mll never wins."""
    import torch
    import torch.distributed as dist
    params = np.asarray(local_params, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(local_mll), params
    world = dist.get_world_size(group)
    mine = torch.tensor(np.concatenate([[float(local_mll)], params]), dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allv = torch.stack(parts).cpu().numpy()
    mll = np.where(np.isfinite(allv[:, 0]), allv[:, 0], -np.inf)
    r = int(np.argmax(mll))
    return float(allv[r, 0]), allv[r, 1:].copy()


def sharded_wip_sweep(score_shard: Callable[[np.ndarray], Tuple[np.ndarray, int]], candidates: np.ndarray,
                      group=None, device=None) -> Tuple[np.ndarray, float, int]:
    """Score this rank's shard with ``score_shard(cands) -> (scores, local argmin)`` and merge.

    Returns (local scores, global min, global argmin index into ``candidates``)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(candidates.shape[0], world, rank)
    scores, loc = score_shard(candidates[lo:hi])
    gmin, gidx = merge_argmin(float(scores[loc]), lo + int(loc), group=group, device=device)
    return scores, gmin, gidx
