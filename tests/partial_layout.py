"""Test helper: a per-throttle `used` laid out as the engine's partial buffer (values, key-presence counts, pod count,
error count per throttle row) and read back from a summed buffer.  The layout is NOT restated here: stride and offsets
come from the library (`kt_partial_layout`, the C-ABI query that returns what the kernels are compiled against —
partial_stride / partial_off_* in csrc/kt_device.h), so a change of the aggregate's buffer breaks the world-size-2 test
instead of passing it by.  The product's exchange is `kt_comm_allreduce_partial` (RCCL inside the engine) or any collective
on the buffer `kt_partial_used_buffer` exposes; integer sums are associative, so the result is bit-identical for any world
size."""
from __future__ import annotations

import numpy as np

from kube_throttler_amd import engine as E


def layout(D: int) -> dict:
    return E.partial_layout(D)


def partial_stride(D: int) -> int:
    return layout(D)["stride"]


def pack_partial(used_v, used_present, used_count, error, D: int) -> np.ndarray:
    """Lay a per-throttle `used` (dense rows) out as the engine's partial buffer (what kt_aggregate leaves for one rank;
    presence travels as counts so that it can be summed)."""
    lo = layout(D)
    T = len(used_count)
    out = np.zeros((T, lo["stride"]), dtype=np.int64)
    out[:, lo["values"]:lo["values"] + D] = used_v[:T]
    for d in range(D):
        out[:, lo["presence"] + d] = (np.asarray(used_present[:T]) >> d) & 1
    out[:, lo["pods"]] = used_count[:T]
    out[:, lo["errors"]] = np.asarray(error[:T]) != 0
    return out


def unpack_partial(buf: np.ndarray, D: int):
    """-> (v[T][D], present[T], count[T], has_count[T], error[T]) exactly as kt_finalize reads it."""
    lo = layout(D)
    buf = np.asarray(buf).reshape(-1, lo["stride"])
    vals, pres = buf[:, lo["values"]:lo["values"] + D], buf[:, lo["presence"]:lo["presence"] + D]
    present = np.zeros(len(buf), dtype=np.uint32)
    for d in range(D):
        # present = presence count != 0 or sum != 0 (kt_finalize's rule: the aggregate skips the presence increment for
        # positive values)
        present |= (((pres[:, d] > 0) | (vals[:, d] != 0)).astype(np.uint32) << np.uint32(d))
    v = np.where((pres > 0) | (vals != 0), vals, 0)
    count = buf[:, lo["pods"]]
    return v, present, count, count > 0, buf[:, lo["errors"]] > 0
