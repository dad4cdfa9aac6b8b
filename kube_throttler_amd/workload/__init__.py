"""Synthetic workload generator (C, splitmix64) -> :class:`kube_throttler_amd.snapshot.Snapshot`.

``preset(i)`` returns BASELINE.json ``configs[i]`` (i = 1..4); ``generate(cfg)`` materialises it (or a
pod shard of it) as numpy views over the generator's buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .. import snapshot as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class WorkloadCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_pods_total", C.c_int64), ("pod_begin", C.c_int64), ("n_pods", C.c_int64),
                ("n_thr", C.c_int32), ("n_cluster", C.c_int32), ("D", C.c_int32), ("n_ns", C.c_int32),
                ("K", C.c_int32), ("V", C.c_int32), ("L", C.c_int32), ("terms_min", C.c_int32),
                ("terms_max", C.c_int32), ("reqs_min", C.c_int32), ("reqs_max", C.c_int32), ("rich_ops", C.c_int32),
                ("overrides", C.c_int32), ("now_s", C.c_int64), ("n_invalid_pod_sel", C.c_int32),
                ("n_invalid_ns_sel", C.c_int32), ("n_missing_ns", C.c_int32)]

    def shard(self, rank: int, world: int) -> "WorkloadCfg":
        """Contiguous pod rows of ``rank`` out of ``world`` (row-sharding of SURVEY.md 8e)."""
        c = WorkloadCfg.from_buffer_copy(self)
        per = (self.n_pods_total + world - 1) // world
        c.pod_begin = min(rank * per, self.n_pods_total)
        c.n_pods = max(0, min(per, self.n_pods_total - c.pod_begin))
        return c


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libkt_workload.so")
    srcs = [os.path.join(_HERE, "kt_workload.c"), os.path.join(_HERE, "kt_workload.h"),
            os.path.join(_HERE, "..", "..", "include", "kt_snapshot.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-Wall", "-o", so, srcs[0], "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.kt_workload_preset.argtypes = [C.c_int, C.POINTER(WorkloadCfg)]
        _LIB.kt_workload_generate.restype = C.POINTER(S.KtSnapshot)
        _LIB.kt_workload_generate.argtypes = [C.POINTER(WorkloadCfg)]
        _LIB.kt_workload_free.argtypes = [C.POINTER(S.KtSnapshot)]
    return _LIB


def preset(index: int) -> WorkloadCfg:
    cfg = WorkloadCfg()
    if lib().kt_workload_preset(index, C.byref(cfg)) != 0:
        raise ValueError(f"no preset {index}")
    return cfg


class _Owner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if self.ptr:
            lib().kt_workload_free(self.ptr)
            self.ptr = None


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape)


def _amounts(a: S.KtAmounts, n: int, D: int) -> S.Amounts:
    out = S.Amounts.__new__(S.Amounts)
    out.n, out.D = n, D
    m = max(n, 1)
    out.v = _view(a.v, (m, D), np.int64)
    out.present = _view(a.present, (m,), np.uint32)
    out.count = _view(a.count, (m,), np.int64)
    out.has_count = _view(a.has_count, (m,), np.uint8)
    return out


def _reqs(r: S.KtReqs) -> S.ReqsArrays:
    n = int(r.n)
    off = _view(r.val_off, (n + 1,), np.uint32)
    nv = int(off[n]) if n else 0
    return S.ReqsArrays(_view(r.op, (n,), np.uint8), _view(r.key, (n,), np.uint32), off, _view(r.val, (nv,), np.uint32))


def generate(cfg: WorkloadCfg) -> S.Snapshot:
    ptr = lib().kt_workload_generate(C.byref(cfg))
    if not ptr:
        raise ValueError("invalid workload configuration")
    k = ptr.contents
    s = S.Snapshot(int(k.D), int(k.L))
    s._owner = _Owner(ptr)
    s.cfg = cfg
    n_ns, n, T, D = int(k.n_ns), int(k.n_pods), int(k.n_thr), int(k.D)
    s.n_ns, s.n_pods, s.n_thr = n_ns, n, T
    s.ns_valid = _view(k.ns_valid, (max(n_ns, 1),), np.uint8)
    s.ns_label_off = _view(k.ns_label_off, (n_ns + 1,), np.uint32)
    nl = int(s.ns_label_off[n_ns])
    s.ns_label_key = _view(k.ns_label_key, (max(nl, 1),), np.uint32)
    s.ns_label_pair = _view(k.ns_label_pair, (max(nl, 1),), np.uint32)
    m = max(n, 1)
    s.pod_ns = _view(k.pod_ns, (m,), np.uint32)
    s.pod_flags = _view(k.pod_flags, (m,), np.uint32)
    s.pod_label_off = _view(k.pod_label_off, (n + 1,), np.uint32)
    nl = int(s.pod_label_off[n])
    s.pod_label_key = _view(k.pod_label_key, (max(nl, 1),), np.uint32)
    s.pod_label_pair = _view(k.pod_label_pair, (max(nl, 1),), np.uint32)
    s.pod_ctr_off = _view(k.pod_ctr_off, (n + 1,), np.uint32)
    nc = int(s.pod_ctr_off[n])
    s.ctr_init = _view(k.ctr_init, (max(nc, 1),), np.uint8)
    s.ctr_present = _view(k.ctr_present, (max(nc, 1),), np.uint32)
    s.ctr_req = _view(k.ctr_req, (max(nc, 1), D), np.int64)
    s.pod_ovh_present = _view(k.pod_ovh_present, (m,), np.uint32)
    s.pod_ovh = _view(k.pod_ovh, (m, D), np.int64)
    mt = max(T, 1)
    s.thr_flags = _view(k.thr_flags, (mt,), np.uint32)
    s.thr_ns = _view(k.thr_ns, (mt,), np.uint32)
    s.thr_spec, s.thr_calc = _amounts(k.thr_spec, T, D), _amounts(k.thr_calc, T, D)
    s.thr_used, s.thr_reserved = _amounts(k.thr_used, T, D), _amounts(k.thr_reserved, T, D)
    s.thr_thrl_flag = _view(k.thr_thrl_flag, (mt,), np.uint32)
    s.thr_thrl_has = _view(k.thr_thrl_has, (mt,), np.uint32)
    s.thr_status_msgs_fp = _view(k.thr_status_msgs_fp, (mt,), np.uint64)
    s.thr_spec_msgs_fp = _view(k.thr_spec_msgs_fp, (mt,), np.uint64)
    s.thr_ovr_off = _view(k.thr_ovr_off, (T + 1,), np.uint32)
    no = int(s.thr_ovr_off[T])
    mo = max(no, 1)
    s.ovr_begin_s = _view(k.ovr_begin_s, (mo,), np.int64)
    s.ovr_begin_ns = _view(k.ovr_begin_ns, (mo,), np.int32)
    s.ovr_end_s = _view(k.ovr_end_s, (mo,), np.int64)
    s.ovr_end_ns = _view(k.ovr_end_ns, (mo,), np.int32)
    s.ovr_flags = _view(k.ovr_flags, (mo,), np.uint8)
    s.ovr_thr = _amounts(k.ovr_thr, no, D)
    s.thr_term_off = _view(k.thr_term_off, (T + 1,), np.uint32)
    nt = int(s.thr_term_off[T])
    s.term_flags = _view(k.term_flags, (max(nt, 1),), np.uint8)
    s.term_preq_off = _view(k.term_preq_off, (nt + 1,), np.uint32)
    s.term_nreq_off = _view(k.term_nreq_off, (nt + 1,), np.uint32)
    s.preq, s.nreq = _reqs(k.preq), _reqs(k.nreq)
    return s


def small(seed=1, n_pods=2000, n_thr=64, n_cluster=32, D=8, n_ns=8, K=8, V=4, L=4, terms=(1, 3), reqs=(0, 3),
          rich_ops=1, overrides=1, n_invalid_pod_sel=0, n_invalid_ns_sel=0, n_missing_ns=0) -> WorkloadCfg:
    """A dense little configuration for parity tests (high match rates, every selector feature)."""
    c = WorkloadCfg()
    c.seed, c.n_pods_total, c.pod_begin, c.n_pods = seed, n_pods, 0, n_pods
    c.n_thr, c.n_cluster, c.D, c.n_ns, c.K, c.V, c.L = n_thr, n_cluster, D, n_ns, K, V, L
    c.terms_min, c.terms_max, c.reqs_min, c.reqs_max = terms[0], terms[1], reqs[0], reqs[1]
    c.rich_ops, c.overrides, c.now_s = rich_ops, overrides, 1767225600
    c.n_invalid_pod_sel, c.n_invalid_ns_sel, c.n_missing_ns = n_invalid_pod_sel, n_invalid_ns_sel, n_missing_ns
    return c
