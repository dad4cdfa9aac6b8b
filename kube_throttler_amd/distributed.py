"""Row-sharding helpers for multi-GPU runs (one process per GPU, torch.distributed: nccl = RCCL on ROCm).

The hot path shards by pods: rank r holds the contiguous pod rows [r*P/N, (r+1)*P/N) and a full replica of
the throttle tables.  The only exchange is one sum all-reduce of the int64 partial-`used` buffer
``[T][2D+2]`` (values, key-presence counts, pod count, error count) between ``kt_aggregate`` and
``kt_finalize`` — integer sums are associative, so the result is bit-identical for any world size.
"""
from __future__ import annotations

import numpy as np


def partial_stride(D: int) -> int:
    return 2 * D + 2


def pack_partial(used_v, used_present, used_count, error, D: int) -> np.ndarray:
    """Lay a per-throttle `used` (dense rows) out as the engine's partial buffer (host-side twin of the
    kt_aggregate output; presence travels as counts so that it can be summed)."""
    T = len(used_count)
    out = np.zeros((T, partial_stride(D)), dtype=np.int64)
    out[:, :D] = used_v[:T]
    for d in range(D):
        out[:, D + d] = (np.asarray(used_present[:T]) >> d) & 1
    out[:, 2 * D] = used_count[:T]
    out[:, 2 * D + 1] = np.asarray(error[:T]) != 0
    return out


def unpack_partial(buf: np.ndarray, D: int):
    """-> (v[T][D], present[T], count[T], has_count[T], error[T]) exactly as kt_finalize reads it."""
    buf = np.asarray(buf).reshape(-1, partial_stride(D))
    present = np.zeros(len(buf), dtype=np.uint32)
    for d in range(D):
        # present = presence count != 0 or sum != 0 (kt_finalize's rule: the L2-form aggregate skips the presence
        # increment for positive values)
        present |= (((buf[:, D + d] > 0) | (buf[:, d] != 0)).astype(np.uint32) << np.uint32(d))
    v = np.where((buf[:, D:2 * D] > 0) | (buf[:, :D] != 0), buf[:, :D], 0)
    count = buf[:, 2 * D]
    return v, present, count, count > 0, buf[:, 2 * D + 1] > 0


def allreduce_partial(tensor, dist):
    """One collective per reconcile: sum over ranks, in place."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
