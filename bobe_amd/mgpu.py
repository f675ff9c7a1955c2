"""The library's own RCCL exchange step (include/bobe_gp.h ``bobe_mgpu_*``), one process per GPU.

``dist_sweep`` does the same merge with ``torch.distributed`` collectives; here the all-gather of the per-shard
(min score, global index) pairs and of the restarts' (mll, theta) is issued by libbobe_gp.so itself on a RCCL
communicator it owns — what a host program that is not Python (or has no torch) binds.  The 128-byte RCCL unique id
has to reach every rank through some channel of the host program; ``init_from_torch`` uses the process group that is
already there, ``init`` takes the bytes from anywhere."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

ID_BYTES = 128


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    _lib.check(_lib.load().bobe_mgpu_unique_id(buf), "bobe_mgpu_unique_id")
    return buf.raw


def init(uid: bytes, world: int, rank: int, device: int) -> None:
    if len(uid) != ID_BYTES:
        raise ValueError("the RCCL unique id has 128 bytes")
    _lib.check(_lib.load().bobe_mgpu_init(C.create_string_buffer(uid, ID_BYTES), int(world), int(rank), int(device)),
               "bobe_mgpu_init")


def init_from_torch(device: int, group=None) -> Tuple[int, int]:
    """Rank 0 draws the unique id, the initialised ``torch.distributed`` group carries it to the others."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    else:
        world, rank, uid = 1, 0, unique_id()
    init(uid, world, rank, device)
    return world, rank


def world() -> int:
    return int(_lib.load().bobe_mgpu_world())


def finalize() -> None:
    _lib.load().bobe_mgpu_finalize()


def wip_sweep(gp, cand_shard, global_offset: int, mc_points, want_scores: bool = True) -> dict:
    """``GP.wip_sweep`` on this rank's shard + the all-gather merge: dict(wipv, wipstd (local shard), argmin_v, min_v,
    argmin_s, min_s (global, indices into the whole candidate set))."""
    dev = hasattr(cand_shard, "data_ptr")
    cand = cand_shard if dev else _lib.as_f64(np.atleast_2d(cand_shard))
    z = mc_points if hasattr(mc_points, "data_ptr") else _lib.as_f64(np.atleast_2d(mc_points))
    c, m = int(cand.shape[0]), int(z.shape[0])
    wipv = np.empty(c) if want_scores else None
    wipstd = np.empty(c) if want_scores else None
    av, asd, mv, ms = C.c_int64(-1), C.c_int64(-1), C.c_double(0.0), C.c_double(0.0)
    _lib.check(gp._lib.bobe_mgpu_wip_sweep(gp._h, _lib.ptr(cand) if c else None, c, int(global_offset), _lib.ptr(z), m,
                                           float(gp.y_std), _lib.ptr(wipv), _lib.ptr(wipstd), None, None,
                                           C.byref(av), C.byref(mv), C.byref(asd), C.byref(ms)), "bobe_mgpu_wip_sweep")
    return {"wipv": wipv, "wipstd": wipstd, "argmin_v": av.value, "min_v": mv.value, "argmin_s": asd.value,
            "min_s": ms.value}


def best_fit(mll: float, theta) -> Tuple[float, np.ndarray]:
    th = _lib.as_f64(theta).reshape(-1)
    out = np.empty_like(th)
    best = C.c_double(0.0)
    _lib.check(_lib.load().bobe_mgpu_best_fit(float(mll), _lib.ptr(th), th.size, C.byref(best), _lib.ptr(out)),
               "bobe_mgpu_best_fit")
    return best.value, out
