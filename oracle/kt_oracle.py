"""ctypes wrapper for the CPU oracle (oracle/kt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (kube_throttler_amd) must never do so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from kube_throttler_amd import snapshot as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libkt_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("kt_oracle.c", "kt_oracle.h")] + \
        [os.path.join(_HERE, "..", "include", "kt_snapshot.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class _ReconcileOut(C.Structure):
    _fields_ = [("used", S.KtAmounts), ("calc", S.KtAmounts), ("calc_updated", C.POINTER(C.c_uint8)),
                ("thrl_flag", C.POINTER(C.c_uint32)), ("thrl_has", C.POINTER(C.c_uint32)),
                ("thrl_pod", C.POINTER(C.c_uint8)), ("error", C.POINTER(C.c_uint8))]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.kto_create.restype = C.c_void_p
        _LIB.kto_create.argtypes = [C.POINTER(S.KtSnapshot)]
        _LIB.kto_destroy.argtypes = [C.c_void_p]
        _LIB.kto_enable_ns_memo.argtypes = [C.c_void_p]
        _LIB.kto_check.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _LIB.kto_admit.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(S.KtAmounts)]
        _LIB.kto_next_override.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.kto_pod_requests.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.kto_reconcile.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                       C.POINTER(_ReconcileOut), C.c_int]
        _LIB.kto_set_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _LIB


class ReconcileResult:
    def __init__(self, n, D):
        self.used, self.calc = S.Amounts(n, D), S.Amounts(n, D)
        m = max(n, 1)
        self.calc_updated = np.zeros(m, np.uint8)
        self.thrl_flag = np.zeros(m, np.uint32)
        self.thrl_has = np.zeros(m, np.uint32)
        self.thrl_pod = np.zeros(m, np.uint8)
        self.error = np.zeros(m, np.uint8)


def effective_cpus() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container with
    cpu.max = 16 CPUs on a 256-thread box runs 256 OpenMP threads SLOWER than 16: measured 4.5 s against 2.9 s for the same
    check).  What the tests hand to `nthreads` and what bench.py reports as `cpu_baseline.cores`."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and pr > 0:
                n = min(n, max(1, q // pr))
        except Exception:
            pass
    return max(1, n)


class Oracle:
    def __init__(self, snap: S.Snapshot, memo: bool = True):
        """memo: test mode — the namespace side of every ClusterThrottle term is evaluated once per (term, namespace)
        instead of once per (term, pod) (kto_enable_ns_memo: same results; check(mimic_log_args=True), the timed CPU
        baseline, never uses it)."""
        self.snap = snap
        self._struct = snap.as_struct()
        self._ctx = lib().kto_create(C.byref(self._struct))
        if memo:
            lib().kto_enable_ns_memo(self._ctx)

    def close(self):
        if self._ctx:
            lib().kto_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def refresh(self):
        """Re-point at the snapshot arrays after status arrays were mutated in place (no re-index needed)."""
        return self

    def pod_requests(self, rows=None):
        n = self.snap.n_pods if rows is None else len(rows)
        rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        v = np.zeros((max(n, 1), self.snap.D), np.int64)
        present = np.zeros(max(n, 1), np.uint32)
        lib().kto_pod_requests(self._ctx, n, None if rows_a is None else rows_a.ctypes.data, v.ctypes.data,
                               present.ctypes.data)
        return v[:n], present[:n]

    def check(self, rows=None, on_equal=False, want_status=True, nthreads=1, mimic_log_args=False):
        n = self.snap.n_pods if rows is None else len(rows)
        rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        T = self.snap.n_thr
        status = np.zeros((max(n, 1), max(T, 1)), np.uint8) if want_status else None
        summary = np.zeros(max(n, 1), np.uint64)
        lib().kto_check(self._ctx, n, None if rows_a is None else rows_a.ctypes.data, int(on_equal),
                        None if status is None else status.ctypes.data, summary.ctypes.data, nthreads,
                        int(mimic_log_args))
        return (None if status is None else status[:n, :T]), summary[:n]

    def next_override(self, now=(0, 0)):
        """NextOverrideHappensIn of every throttle as (instant seconds, nanoseconds, has)."""
        T = self.snap.n_thr
        sec, nsec, has = np.zeros(max(T, 1), np.int64), np.zeros(max(T, 1), np.int32), np.zeros(max(T, 1), np.uint8)
        lib().kto_next_override(self._ctx, T, None, int(now[0]), int(now[1]), sec.ctypes.data, nsec.ctypes.data,
                                has.ctypes.data)
        return sec[:T], nsec[:T], has[:T]

    def admit(self, rows=None, on_equal=False):
        """One in-order scheduling pass: PreFilter, and on Success Reserve.  -> (status, summary, reserved totals)."""
        n = self.snap.n_pods if rows is None else len(rows)
        rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        T = self.snap.n_thr
        status = np.zeros((max(n, 1), max(T, 1)), np.uint8)
        summary = np.zeros(max(n, 1), np.uint64)
        reserved = S.Amounts(T, self.snap.D)
        st = reserved.as_struct()
        lib().kto_admit(self._ctx, n, None if rows_a is None else rows_a.ctypes.data, int(on_equal),
                        status.ctypes.data, summary.ctypes.data, C.byref(st))
        return status[:n, :T], summary[:n], reserved

    def set_status_used_hi(self, hi):
        """status.used beyond int64 (resource.Quantity never overflows): hi [n_thr][D] = the high 64 bits of the snapshot's
        thr_used.v values; None = every value is the int64 the snapshot holds."""
        self._status_hi = None if hi is None else np.ascontiguousarray(hi, dtype=np.int64)
        lib().kto_set_wide(self._ctx, None if self._status_hi is None else self._status_hi.ctypes.data, None)

    def reconcile(self, now=(0, 0), rows=None, nthreads=1, wide=False) -> ReconcileResult:
        """wide: `used` as 128-bit values — result.used.v holds the low words, result.used_hi the high words."""
        n = self.snap.n_thr if rows is None else len(rows)
        rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        r = ReconcileResult(n, self.snap.D)
        st_hi = getattr(self, "_status_hi", None)
        if wide:
            r.used_hi = np.zeros((max(n, 1), self.snap.D), np.int64)
            lib().kto_set_wide(self._ctx, None if st_hi is None else st_hi.ctypes.data, r.used_hi.ctypes.data)
        out = _ReconcileOut(r.used.as_struct(), r.calc.as_struct(),
                            r.calc_updated.ctypes.data_as(C.POINTER(C.c_uint8)),
                            r.thrl_flag.ctypes.data_as(C.POINTER(C.c_uint32)),
                            r.thrl_has.ctypes.data_as(C.POINTER(C.c_uint32)),
                            r.thrl_pod.ctypes.data_as(C.POINTER(C.c_uint8)),
                            r.error.ctypes.data_as(C.POINTER(C.c_uint8)))
        lib().kto_reconcile(self._ctx, n, None if rows_a is None else rows_a.ctypes.data, int(now[0]), int(now[1]),
                            C.byref(out), nthreads)
        if wide:
            lib().kto_set_wide(self._ctx, None if st_hi is None else st_hi.ctypes.data, None)
        return r


# ---- function-level entry points (reference unit-test tables) -------------------------------------
def unit_is_throttled(D, threshold: S.Amounts, used: S.Amounts, on_equal: bool):
    """ResourceAmount.IsThrottled on row 0 of each table -> (pod, {dim: bool})."""
    L = lib()
    flag, has, pod = C.c_uint32(0), C.c_uint32(0), C.c_uint8(0)
    ts, us = threshold.as_struct(), used.as_struct()
    L.kto_unit_is_throttled(D, C.byref(ts), C.byref(us), int(on_equal), C.byref(flag), C.byref(has), C.byref(pod))
    return bool(pod.value), {d: bool(flag.value >> d & 1) for d in range(D) if has.value >> d & 1}


def unit_is_throttled_for(snap: S.Snapshot, pod_row: int, flags: dict, pod_flag: bool) -> bool:
    st = snap.as_struct()
    has = sum(1 << d for d in flags)
    flag = sum(1 << d for d, v in flags.items() if v)
    return bool(lib().kto_unit_is_throttled_for(C.byref(st), C.c_int64(pod_row), C.c_uint32(flag), C.c_uint32(has),
                                                int(pod_flag)))


def unit_override_is_active(snap: S.Snapshot, o: int, now):
    st = snap.as_struct()
    return lib().kto_unit_override_is_active(C.byref(st), C.c_uint32(o), C.c_int64(now[0]), C.c_int32(now[1]))


def unit_calculate_threshold(snap: S.Snapshot, t: int, now):
    st = snap.as_struct()
    out = S.Amounts(1, snap.D)
    os_ = out.as_struct()
    any_err = C.c_uint8(0)
    rc = lib().kto_unit_calculate_threshold(C.byref(st), C.c_int32(t), C.c_int64(now[0]), C.c_int32(now[1]),
                                            C.byref(os_), C.byref(any_err))
    assert rc == 0
    return out, bool(any_err.value)


def unit_selector_matches(snap: S.Snapshot, t: int, pod_row: int) -> int:
    st = snap.as_struct()
    return lib().kto_unit_selector_matches(C.byref(st), C.c_int32(t), C.c_int64(pod_row))
