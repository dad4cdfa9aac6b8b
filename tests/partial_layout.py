"""Test helper: the layout of the engine's partial-`used` buffer ``[T][2D+2]`` (values, key-presence counts, pod count,
error count) on the host — how the oracle's per-shard `used` is laid out as the buffer ranks exchange, and how a summed
buffer reads back.  The product's exchange is `kt_comm_allreduce_partial` (RCCL inside the engine) or any collective on
the buffer `kt_partial_used_buffer` exposes; integer sums are associative, so the result is bit-identical for any world
size."""
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
