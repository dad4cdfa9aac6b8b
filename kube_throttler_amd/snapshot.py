"""ctypes mirror of include/kt_snapshot.h plus a numpy-backed container.

A :class:`Snapshot` owns numpy arrays for every pointer in ``kt_snapshot`` and hands out a ctypes
struct whose pointers alias them (the arrays must outlive any native call that uses the struct).
It is the single flat format shared by the workload generator, the engine's bulk ingest and the
CPU oracle (tests only).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

KT_MAX_DIMS = 16
KT_MAX_LABELS = 64

POD_VALID, POD_SCHED_MATCH, POD_SCHEDULED, POD_FINISHED = 0x1, 0x2, 0x4, 0x8
THR_VALID, THR_CLUSTER, THR_RESPONSIBLE, THR_CALC_AT_NONZERO, THR_THROTTLED_POD = 0x1, 0x2, 0x4, 0x8, 0x10
TERM_POD_SEL_INVALID, TERM_NS_SEL_INVALID = 0x1, 0x2
OP_IN, OP_NOT_IN, OP_EXISTS, OP_DOES_NOT_EXIST = 0, 1, 2, 3
OVR_PARSE_ERROR = 0x1
OVR_BEGIN_PARSED = 0x2
ZERO_TIME_S = -62135596800

# per (pod, throttle) status codes / per-pod verdicts (include/kt_engine.h)
NOT_AFFECTED, NOT_THROTTLED, ACTIVE, INSUFFICIENT, EXCEEDS, ERROR = 0, 1, 2, 3, 4, 255
VERDICT_ALLOW, VERDICT_BLOCK, VERDICT_ERROR = 0, 1, 2
STATUS_NAMES = {
    NOT_AFFECTED: "not-affected",
    NOT_THROTTLED: "not-throttled",
    ACTIVE: "active",
    INSUFFICIENT: "insufficient",
    EXCEEDS: "pod-requests-exceeds-threshold",
    ERROR: "error",
}

_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class KtAmounts(C.Structure):
    _fields_ = [("v", _i64p), ("present", _u32p), ("count", _i64p), ("has_count", _u8p)]


class KtReqs(C.Structure):
    _fields_ = [("n", C.c_uint32), ("op", _u8p), ("key", _u32p), ("val_off", _u32p), ("val", _u32p)]


class KtSnapshot(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("L", C.c_int32),
        ("n_ns", C.c_int32), ("ns_valid", _u8p), ("ns_label_off", _u32p), ("ns_label_key", _u32p),
        ("ns_label_pair", _u32p),
        ("n_pods", C.c_int64), ("pod_ns", _u32p), ("pod_flags", _u32p), ("pod_label_off", _u32p),
        ("pod_label_key", _u32p), ("pod_label_pair", _u32p), ("pod_ctr_off", _u32p), ("ctr_init", _u8p),
        ("ctr_present", _u32p), ("ctr_req", _i64p), ("pod_ovh_present", _u32p), ("pod_ovh", _i64p),
        ("n_thr", C.c_int32), ("thr_flags", _u32p), ("thr_ns", _u32p),
        ("thr_spec", KtAmounts), ("thr_calc", KtAmounts), ("thr_used", KtAmounts), ("thr_reserved", KtAmounts),
        ("thr_thrl_flag", _u32p), ("thr_thrl_has", _u32p), ("thr_status_msgs_fp", _u64p),
        ("thr_spec_msgs_fp", _u64p),
        ("thr_ovr_off", _u32p), ("ovr_begin_s", _i64p), ("ovr_begin_ns", _i32p), ("ovr_end_s", _i64p),
        ("ovr_end_ns", _i32p), ("ovr_flags", _u8p), ("ovr_thr", KtAmounts),
        ("thr_term_off", _u32p), ("term_flags", _u8p), ("term_preq_off", _u32p), ("term_nreq_off", _u32p),
        ("preq", KtReqs), ("nreq", KtReqs),
    ]


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Amounts:
    """n rows of dense ResourceAmount (v[n][D], present, count, has_count)."""

    def __init__(self, n: int, D: int):
        self.n, self.D = n, D
        self.v = np.zeros((max(n, 1), D), dtype=np.int64)
        self.present = np.zeros(max(n, 1), dtype=np.uint32)
        self.count = np.zeros(max(n, 1), dtype=np.int64)
        self.has_count = np.zeros(max(n, 1), dtype=np.uint8)

    def as_struct(self) -> KtAmounts:
        return KtAmounts(_ptr(self.v, C.c_int64), _ptr(self.present, C.c_uint32), _ptr(self.count, C.c_int64),
                         _ptr(self.has_count, C.c_uint8))

    def set_row(self, i, values: dict, count=None):
        """values: {dim: int}; count None => resourceCounts nil."""
        self.v[i, :] = 0
        self.present[i] = 0
        for d, x in values.items():
            self.v[i, d] = x
            self.present[i] |= np.uint32(1 << d)
        self.has_count[i] = 0 if count is None else 1
        self.count[i] = 0 if count is None else count

    def copy(self) -> "Amounts":
        o = Amounts(self.n, self.D)
        o.v[...] = self.v
        o.present[...] = self.present
        o.count[...] = self.count
        o.has_count[...] = self.has_count
        return o


class Reqs:
    """CSR pool of label-selector requirements."""

    def __init__(self):
        self.op: list[int] = []
        self.key: list[int] = []
        self.val_off: list[int] = [0]
        self.val: list[int] = []
        self._arrs = None

    def add(self, op: int, key: int, vals=()):
        self.op.append(op)
        self.key.append(key)
        self.val.extend(vals)
        self.val_off.append(len(self.val))
        self._arrs = None

    def __len__(self):
        return len(self.op)

    def arrays(self):
        if self._arrs is None:
            self._arrs = (
                np.asarray(self.op + [0], dtype=np.uint8),
                np.asarray(self.key + [0], dtype=np.uint32),
                np.asarray(self.val_off, dtype=np.uint32),
                np.asarray(self.val + [0], dtype=np.uint32),
            )
        return self._arrs

    def as_struct(self) -> KtReqs:
        op, key, off, val = self.arrays()
        return KtReqs(len(self.op), _ptr(op, C.c_uint8), _ptr(key, C.c_uint32), _ptr(off, C.c_uint32),
                      _ptr(val, C.c_uint32))

    @classmethod
    def from_arrays(cls, op, key, val_off, val):
        r = cls()
        r.op = list(map(int, op))
        r.key = list(map(int, key))
        r.val_off = list(map(int, val_off))
        r.val = list(map(int, val))
        return r


class ReqsArrays:
    """Requirement pool backed directly by numpy arrays (generator output)."""

    def __init__(self, op, key, val_off, val):
        self.n = len(op)
        self.op = np.ascontiguousarray(np.append(op, 0).astype(np.uint8))
        self.key = np.ascontiguousarray(np.append(key, 0).astype(np.uint32))
        self.val_off = np.ascontiguousarray(np.asarray(val_off, dtype=np.uint32))
        self.val = np.ascontiguousarray(np.append(val, 0).astype(np.uint32))

    def __len__(self):
        return self.n

    def as_struct(self) -> KtReqs:
        return KtReqs(self.n, _ptr(self.op, C.c_uint8), _ptr(self.key, C.c_uint32), _ptr(self.val_off, C.c_uint32),
                      _ptr(self.val, C.c_uint32))


class Snapshot:
    """numpy-backed kt_snapshot.  Attribute names follow the C struct."""

    def __init__(self, D: int, L: int):
        assert 1 <= D <= KT_MAX_DIMS and 0 <= L <= KT_MAX_LABELS
        self.D, self.L = D, L
        self.n_ns = 0
        self.n_pods = 0
        self.n_thr = 0
        self._keep = None

    # -- allocation helpers -------------------------------------------------------------------
    def alloc_namespaces(self, n_ns: int, n_labels: int):
        self.n_ns = n_ns
        self.ns_valid = np.ones(max(n_ns, 1), dtype=np.uint8)
        self.ns_label_off = np.zeros(n_ns + 1, dtype=np.uint32)
        self.ns_label_key = np.zeros(max(n_labels, 1), dtype=np.uint32)
        self.ns_label_pair = np.zeros(max(n_labels, 1), dtype=np.uint32)

    def alloc_pods(self, n_pods: int, n_labels: int, n_ctr: int):
        D = self.D
        self.n_pods = n_pods
        m = max(n_pods, 1)
        self.pod_ns = np.zeros(m, dtype=np.uint32)
        self.pod_flags = np.zeros(m, dtype=np.uint32)
        self.pod_label_off = np.zeros(n_pods + 1, dtype=np.uint32)
        self.pod_label_key = np.zeros(max(n_labels, 1), dtype=np.uint32)
        self.pod_label_pair = np.zeros(max(n_labels, 1), dtype=np.uint32)
        self.pod_ctr_off = np.zeros(n_pods + 1, dtype=np.uint32)
        self.ctr_init = np.zeros(max(n_ctr, 1), dtype=np.uint8)
        self.ctr_present = np.zeros(max(n_ctr, 1), dtype=np.uint32)
        self.ctr_req = np.zeros((max(n_ctr, 1), D), dtype=np.int64)
        self.pod_ovh_present = np.zeros(m, dtype=np.uint32)
        self.pod_ovh = np.zeros((m, D), dtype=np.int64)

    def alloc_throttles(self, n_thr: int, n_ovr: int, n_term: int):
        D = self.D
        self.n_thr = n_thr
        m = max(n_thr, 1)
        self.thr_flags = np.zeros(m, dtype=np.uint32)
        self.thr_ns = np.zeros(m, dtype=np.uint32)
        self.thr_spec = Amounts(n_thr, D)
        self.thr_calc = Amounts(n_thr, D)
        self.thr_used = Amounts(n_thr, D)
        self.thr_reserved = Amounts(n_thr, D)
        self.thr_thrl_flag = np.zeros(m, dtype=np.uint32)
        self.thr_thrl_has = np.zeros(m, dtype=np.uint32)
        self.thr_status_msgs_fp = np.zeros(m, dtype=np.uint64)
        self.thr_spec_msgs_fp = np.zeros(m, dtype=np.uint64)
        self.thr_ovr_off = np.zeros(n_thr + 1, dtype=np.uint32)
        k = max(n_ovr, 1)
        self.ovr_begin_s = np.full(k, ZERO_TIME_S, dtype=np.int64)
        self.ovr_begin_ns = np.zeros(k, dtype=np.int32)
        self.ovr_end_s = np.full(k, ZERO_TIME_S, dtype=np.int64)
        self.ovr_end_ns = np.zeros(k, dtype=np.int32)
        self.ovr_flags = np.zeros(k, dtype=np.uint8)
        self.ovr_thr = Amounts(n_ovr, D)
        self.thr_term_off = np.zeros(n_thr + 1, dtype=np.uint32)
        self.term_flags = np.zeros(max(n_term, 1), dtype=np.uint8)
        self.term_preq_off = np.zeros(n_term + 1, dtype=np.uint32)
        self.term_nreq_off = np.zeros(n_term + 1, dtype=np.uint32)
        self.preq = Reqs()
        self.nreq = Reqs()

    @property
    def n_term(self) -> int:
        return int(self.thr_term_off[self.n_thr])

    @property
    def n_ovr(self) -> int:
        return int(self.thr_ovr_off[self.n_thr])

    # -- ctypes view --------------------------------------------------------------------------
    def as_struct(self) -> KtSnapshot:
        s = KtSnapshot()
        s.D, s.L = self.D, self.L
        s.n_ns = self.n_ns
        s.ns_valid = _ptr(self.ns_valid, C.c_uint8)
        s.ns_label_off = _ptr(self.ns_label_off, C.c_uint32)
        s.ns_label_key = _ptr(self.ns_label_key, C.c_uint32)
        s.ns_label_pair = _ptr(self.ns_label_pair, C.c_uint32)
        s.n_pods = self.n_pods
        for name, ct in (("pod_ns", C.c_uint32), ("pod_flags", C.c_uint32), ("pod_label_off", C.c_uint32),
                         ("pod_label_key", C.c_uint32), ("pod_label_pair", C.c_uint32),
                         ("pod_ctr_off", C.c_uint32), ("ctr_init", C.c_uint8), ("ctr_present", C.c_uint32),
                         ("ctr_req", C.c_int64), ("pod_ovh_present", C.c_uint32), ("pod_ovh", C.c_int64)):
            setattr(s, name, _ptr(getattr(self, name), ct))
        s.n_thr = self.n_thr
        s.thr_flags = _ptr(self.thr_flags, C.c_uint32)
        s.thr_ns = _ptr(self.thr_ns, C.c_uint32)
        s.thr_spec = self.thr_spec.as_struct()
        s.thr_calc = self.thr_calc.as_struct()
        s.thr_used = self.thr_used.as_struct()
        s.thr_reserved = self.thr_reserved.as_struct()
        s.thr_thrl_flag = _ptr(self.thr_thrl_flag, C.c_uint32)
        s.thr_thrl_has = _ptr(self.thr_thrl_has, C.c_uint32)
        s.thr_status_msgs_fp = _ptr(self.thr_status_msgs_fp, C.c_uint64)
        s.thr_spec_msgs_fp = _ptr(self.thr_spec_msgs_fp, C.c_uint64)
        s.thr_ovr_off = _ptr(self.thr_ovr_off, C.c_uint32)
        s.ovr_begin_s = _ptr(self.ovr_begin_s, C.c_int64)
        s.ovr_begin_ns = _ptr(self.ovr_begin_ns, C.c_int32)
        s.ovr_end_s = _ptr(self.ovr_end_s, C.c_int64)
        s.ovr_end_ns = _ptr(self.ovr_end_ns, C.c_int32)
        s.ovr_flags = _ptr(self.ovr_flags, C.c_uint8)
        s.ovr_thr = self.ovr_thr.as_struct()
        s.thr_term_off = _ptr(self.thr_term_off, C.c_uint32)
        s.term_flags = _ptr(self.term_flags, C.c_uint8)
        s.term_preq_off = _ptr(self.term_preq_off, C.c_uint32)
        s.term_nreq_off = _ptr(self.term_nreq_off, C.c_uint32)
        s.preq = self.preq.as_struct()
        s.nreq = self.nreq.as_struct()
        self._keep = s
        return s

    # -- sub-batches: what ONE informer event (or a few) pushes -------------------------------------
    def pod_batch(self, rows) -> "Snapshot":
        """A pods-only batch holding this snapshot's pods `rows`, in that order (for kt_upsert_pods)."""
        rows = np.asarray(rows)
        b = Snapshot(self.D, self.L)
        b.alloc_namespaces(0, 0)
        b.alloc_throttles(0, 0, 0)
        nl = int(sum(int(self.pod_label_off[r + 1]) - int(self.pod_label_off[r]) for r in rows))
        nc = int(sum(int(self.pod_ctr_off[r + 1]) - int(self.pod_ctr_off[r]) for r in rows))
        b.alloc_pods(len(rows), nl, nc)
        lo = co = 0
        for i, r in enumerate(rows):
            b.pod_ns[i], b.pod_flags[i] = self.pod_ns[r], self.pod_flags[r]
            l0, l1 = int(self.pod_label_off[r]), int(self.pod_label_off[r + 1])
            b.pod_label_key[lo:lo + l1 - l0] = self.pod_label_key[l0:l1]
            b.pod_label_pair[lo:lo + l1 - l0] = self.pod_label_pair[l0:l1]
            lo += l1 - l0
            b.pod_label_off[i + 1] = lo
            c0, c1 = int(self.pod_ctr_off[r]), int(self.pod_ctr_off[r + 1])
            b.ctr_init[co:co + c1 - c0] = self.ctr_init[c0:c1]
            b.ctr_present[co:co + c1 - c0] = self.ctr_present[c0:c1]
            b.ctr_req[co:co + c1 - c0] = self.ctr_req[c0:c1]
            co += c1 - c0
            b.pod_ctr_off[i + 1] = co
            b.pod_ovh_present[i] = self.pod_ovh_present[r]
            b.pod_ovh[i] = self.pod_ovh[r]
        return b

    def throttle_batch(self, rows) -> "Snapshot":
        """A throttles-only batch holding this snapshot's throttle rows `rows` (spec, selector, overrides, stored status,
        reserved amounts), for kt_upsert_throttles."""
        rows = np.asarray(rows)
        b = Snapshot(self.D, self.L)
        b.alloc_namespaces(0, 0)
        b.alloc_pods(0, 0, 0)
        n_ovr = int(sum(int(self.thr_ovr_off[t + 1]) - int(self.thr_ovr_off[t]) for t in rows))
        n_term = int(sum(int(self.thr_term_off[t + 1]) - int(self.thr_term_off[t]) for t in rows))
        b.alloc_throttles(len(rows), n_ovr, n_term)
        pools = []
        for pool in (self.preq, self.nreq):
            op, key, val_off, val = pool.arrays() if hasattr(pool, "arrays") else (pool.op, pool.key, pool.val_off, pool.val)
            pools.append((np.asarray(op), np.asarray(key), np.asarray(val_off), np.asarray(val)))
        oo = gg = 0
        for i, t in enumerate(rows):
            b.thr_flags[i], b.thr_ns[i] = self.thr_flags[t], self.thr_ns[t]
            for dst, src in ((b.thr_spec, self.thr_spec), (b.thr_calc, self.thr_calc), (b.thr_used, self.thr_used),
                             (b.thr_reserved, self.thr_reserved)):
                dst.v[i], dst.present[i], dst.count[i], dst.has_count[i] = src.v[t], src.present[t], src.count[t], src.has_count[t]
            b.thr_thrl_flag[i], b.thr_thrl_has[i] = self.thr_thrl_flag[t], self.thr_thrl_has[t]
            b.thr_status_msgs_fp[i], b.thr_spec_msgs_fp[i] = self.thr_status_msgs_fp[t], self.thr_spec_msgs_fp[t]
            for o in range(int(self.thr_ovr_off[t]), int(self.thr_ovr_off[t + 1])):
                b.ovr_begin_s[oo], b.ovr_begin_ns[oo] = self.ovr_begin_s[o], self.ovr_begin_ns[o]
                b.ovr_end_s[oo], b.ovr_end_ns[oo], b.ovr_flags[oo] = self.ovr_end_s[o], self.ovr_end_ns[o], self.ovr_flags[o]
                b.ovr_thr.v[oo], b.ovr_thr.present[oo] = self.ovr_thr.v[o], self.ovr_thr.present[o]
                b.ovr_thr.count[oo], b.ovr_thr.has_count[oo] = self.ovr_thr.count[o], self.ovr_thr.has_count[o]
                oo += 1
            b.thr_ovr_off[i + 1] = oo
            for g in range(int(self.thr_term_off[t]), int(self.thr_term_off[t + 1])):
                b.term_flags[gg] = self.term_flags[g]
                for dst, offs, (op, key, val_off, val) in ((b.preq, self.term_preq_off, pools[0]), (b.nreq, self.term_nreq_off, pools[1])):
                    for r in range(int(offs[g]), int(offs[g + 1])):
                        dst.add(int(op[r]), int(key[r]), [int(v) for v in val[int(val_off[r]):int(val_off[r + 1])]])
                gg += 1
                b.term_preq_off[gg], b.term_nreq_off[gg] = len(b.preq), len(b.nreq)
            b.thr_term_off[i + 1] = gg
        return b

    # -- status write-back (what UpdateStatus would persist) ----------------------------------
    def apply_status(self, used: Amounts, calc: Amounts, calc_updated, thrl_flag, thrl_has, thrl_pod,
                     error=None, rows=None):
        """Store a reconcile result as the new stored CR status (rows: throttle rows of the outputs)."""
        rows = np.arange(self.n_thr) if rows is None else np.asarray(rows)
        for i, t in enumerate(rows):
            if error is not None and error[i]:
                continue
            for dst, src in ((self.thr_used, used), (self.thr_calc, calc)):
                dst.v[t] = src.v[i]
                dst.present[t] = src.present[i]
                dst.count[t] = src.count[i]
                dst.has_count[t] = src.has_count[i]
            self.thr_thrl_flag[t] = thrl_flag[i]
            self.thr_thrl_has[t] = thrl_has[i]
            f = int(self.thr_flags[t]) & ~THR_THROTTLED_POD
            if thrl_pod[i]:
                f |= THR_THROTTLED_POD
            if calc_updated[i]:
                f |= THR_CALC_AT_NONZERO
                self.thr_status_msgs_fp[t] = self.thr_spec_msgs_fp[t] if self._any_parse_error(t) else 0
            self.thr_flags[t] = f

    def _any_parse_error(self, t) -> bool:
        a, b = int(self.thr_ovr_off[t]), int(self.thr_ovr_off[t + 1])
        return bool((self.ovr_flags[a:b] & OVR_PARSE_ERROR).any())


def summary_fields(summary):
    """Decode per-pod summary words -> (verdict, n_exceeds, n_active, n_insufficient)."""
    s = np.asarray(summary, dtype=np.uint64)
    m20 = np.uint64((1 << 20) - 1)
    return (s & np.uint64(3)).astype(np.int64), ((s >> np.uint64(4)) & m20).astype(np.int64), \
        ((s >> np.uint64(24)) & m20).astype(np.int64), ((s >> np.uint64(44)) & m20).astype(np.int64)
