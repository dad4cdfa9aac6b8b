"""ctypes binding of the C-ABI in include/kt_engine.h (libkt_engine.so, hand-written HIP for gfx950).

There is no CPU fallback: a missing library or a box without a GPU raises :class:`EngineError`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import snapshot as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_CSRC, "libkt_engine.so")
# A/B runs only (a fix shown red on the previous library, green on this one): another build of the SAME C-ABI
if os.environ.get("KT_ENGINE_LIB"):
    LIB_PATH = os.path.abspath(os.environ["KT_ENGINE_LIB"])
_LIB = None

KT_OK = 0
ERR_NAMES = {-1: "KT_ERR_INVALID_ARGUMENT", -2: "KT_ERR_OUT_OF_RANGE", -3: "KT_ERR_DEVICE", -4: "KT_ERR_OVERFLOW_RISK",
             -5: "KT_ERR_NOT_READY", -6: "KT_ERR_NO_DEVICE", -7: "KT_ERR_UNSUPPORTED"}
RECONCILE_APPLY = 0x1
ADMIT_COMMIT = 0x1
VARIANT_INCREMENTAL = 0x100  # | VARIANT_INDEXED: pod events keep the `used` partials current (N2)
CHECK_STATUS_MATRIX = 0x1
KERNEL_CHECK, KERNEL_AGGREGATE, KERNEL_FINALIZE, KERNEL_PREPARE, KERNEL_REDUCE = 0, 1, 2, 3, 4
VARIANT_INDEXED, VARIANT_DENSE = 0, 1

EXPORTS = [
    "kt_version", "kt_engine_create", "kt_engine_destroy", "kt_last_error", "kt_upsert_namespaces", "kt_upsert_pods",
    "kt_upsert_throttles", "kt_delete_namespaces", "kt_delete_pods", "kt_delete_throttles", "kt_load_snapshot",
    "kt_set_reserved", "kt_set_status", "kt_reconcile_launch", "kt_aggregate_launch", "kt_partial_used_buffer",
    "kt_use_partial_buffer", "kt_finalize_launch", "kt_reconcile_fetch", "kt_check_launch", "kt_check_fetch", "kt_throttle_rows", "kt_sweep_launch",
    "kt_check_device_summary", "kt_fetch_pod_requests", "kt_timing_enable", "kt_timing_read", "kt_timing_reset",
    "kt_synchronize", "kt_kernel_name", "kt_admit_launch", "kt_fetch_reserved", "kt_reconcile_fetch_next_override",
    "kt_check", "kt_upsert_namespace", "kt_upsert_pod", "kt_upsert_throttle", "kt_comm_unique_id", "kt_comm_init",
    "kt_comm_allreduce_partial", "kt_comm_destroy", "kt_reconcile_rows_launch", "kt_set_exchange_world", "kt_counter", "kt_reconcile_fetch_used_hi",
    "kt_set_wide_sums", "kt_partial_words", "kt_partial_layout", "kt_debug_reload_env", "kt_affected_pods", "kt_paged_check", "kt_paged_reconcile",
]
COUNTER_FEW_CHECKS, COUNTER_COMPILES, COUNTER_INDEX_CHUNKS, COUNTER_INDEX_WORDS, COUNTER_NS_WORD_VISITS, COUNTER_NS_ROWS = range(6)
COUNTER_NS_CHUNK_VISITS, COUNTER_INDEX_IMAGE_WORDS, COUNTER_SLOW_THROTTLES, COUNTER_PACKED_WORDS = 6, 7, 8, 9


def partial_layout(n_dims: int) -> dict:
    """kt_partial_layout(): the layout of one throttle's row of the partial-`used` buffer as the library's kernels are
    compiled (no engine, no GPU needed) -> {stride, values, presence, pods, errors} in int64 words."""
    v = [C.c_int32() for _ in range(5)]
    rc = lib().kt_partial_layout(int(n_dims), *[C.byref(x) for x in v])
    if rc != KT_OK:
        raise EngineError(rc, f"kt_partial_layout({n_dims})")
    return dict(zip(("stride", "values", "presence", "pods", "errors"), (int(x.value) for x in v)))


def version() -> str:
    """kt_version(): library version + hash of the kernel sources it was built from."""
    return lib().kt_version().decode()


def paged_check(engines, n, rows=None, on_equal=False):
    """kt_paged_check: kt_check on every page engine (one per page of <= 16 resource names), combined in the library ->
    (status matrix [n][T], summary words [n])."""
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    T = engines[0].throttle_rows()
    rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    if rows_a is None:
        rows_a = np.arange(n, dtype=np.int64)
    status = np.zeros((max(n, 1), max(T, 1)), np.uint8)
    summary = np.zeros(max(n, 1), np.uint64)
    rc = lib().kt_paged_check(hs, len(engines), n, rows_a.ctypes.data, int(on_equal), summary.ctypes.data, status.ctypes.data)
    if rc != KT_OK:
        raise EngineError(rc, "kt_paged_check: " + "; ".join(lib().kt_last_error(e._h).decode() for e in engines))
    return status[:n, :T], summary[:n]


def paged_reconcile(engines, now, apply=True):
    """kt_paged_reconcile: a reconcile on every page engine -> (per-page ReconcileResult list, replaced_any [T], error_any [T])."""
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    T = engines[0].throttle_rows()
    results = [ReconcileResult(T, e.D) for e in engines]
    structs = (KtStatus * len(engines))(*[r.as_struct() for r in results])
    replaced, err = np.zeros(max(T, 1), np.uint8), np.zeros(max(T, 1), np.uint8)
    rc = lib().kt_paged_reconcile(hs, len(engines), int(now[0]), int(now[1]), RECONCILE_APPLY if apply else 0, T, structs,
                                  replaced.ctypes.data, err.ctypes.data)
    if rc != KT_OK:
        raise EngineError(rc, "kt_paged_reconcile: " + "; ".join(lib().kt_last_error(e._h).decode() for e in engines))
    return results, replaced[:T], err[:T]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class KtConfig(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("max_labels", C.c_int32), ("pod_capacity", C.c_int64),
                ("throttle_capacity", C.c_int32), ("namespace_capacity", C.c_int32), ("device", C.c_int32),
                ("kernel_variant", C.c_int32)]


class KtStatus(C.Structure):
    _fields_ = [("used", S.KtAmounts), ("calc", S.KtAmounts), ("calc_at_nonzero", C.POINTER(C.c_uint8)),
                ("thrl_flag", C.POINTER(C.c_uint32)), ("thrl_has", C.POINTER(C.c_uint32)),
                ("thrl_pod", C.POINTER(C.c_uint8)), ("msgs_fp", C.POINTER(C.c_uint64)),
                ("error", C.POINTER(C.c_uint8))]


def build(force: bool = False) -> str:
    """Compile libkt_engine.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(-6, f"{LIB_PATH} is missing: build it with kube_throttler_amd.engine.build() "
                                  "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.kt_version.restype = C.c_char_p
        L.kt_last_error.restype = C.c_char_p
        L.kt_last_error.argtypes = [C.c_void_p]
        L.kt_kernel_name.restype = C.c_char_p
        L.kt_kernel_name.argtypes = [C.c_void_p, C.c_int32]
        L.kt_engine_create.argtypes = [C.POINTER(KtConfig), C.POINTER(C.c_void_p)]
        L.kt_engine_destroy.argtypes = [C.c_void_p]
        for name in ("kt_upsert_namespaces", "kt_upsert_pods", "kt_upsert_throttles"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(S.KtSnapshot), C.c_void_p]
        L.kt_delete_namespaces.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.kt_delete_pods.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.kt_delete_throttles.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.kt_load_snapshot.argtypes = [C.c_void_p, C.POINTER(S.KtSnapshot)]
        L.kt_set_reserved.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(S.KtAmounts)]
        L.kt_set_status.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(KtStatus)]
        L.kt_reconcile_launch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p]
        L.kt_reconcile_rows_launch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
        L.kt_aggregate_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.kt_partial_used_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.kt_use_partial_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.kt_finalize_launch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p]
        L.kt_reconcile_fetch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(KtStatus)]
        L.kt_check_launch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p]
        L.kt_sweep_launch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p]
        L.kt_check_fetch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.kt_check.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.kt_comm_unique_id.argtypes = [C.c_void_p]
        L.kt_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.kt_comm_allreduce_partial.argtypes = [C.c_void_p, C.c_void_p]
        L.kt_comm_destroy.argtypes = [C.c_void_p]
        L.kt_set_exchange_world.argtypes = [C.c_void_p, C.c_int32]
        L.kt_set_wide_sums.argtypes = [C.c_void_p, C.c_int32]
        L.kt_partial_words.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.kt_partial_layout.argtypes = [C.c_int32] + [C.POINTER(C.c_int32)] * 5
        L.kt_debug_reload_env.argtypes = [C.c_void_p]
        L.kt_affected_pods.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.kt_paged_check.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.kt_paged_reconcile.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32, C.c_uint32, C.c_int32,
                                         C.POINTER(KtStatus), C.c_void_p, C.c_void_p]
        L.kt_counter.argtypes = [C.c_void_p, C.c_int32]
        L.kt_reconcile_fetch_used_hi.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
        L.kt_counter.restype = C.c_int64
        L.kt_throttle_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.kt_check_device_summary.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.kt_fetch_pod_requests.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kt_timing_enable.argtypes = [C.c_void_p, C.c_int32]
        L.kt_timing_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.kt_timing_reset.argtypes = [C.c_void_p]
        L.kt_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        L.kt_reconcile_fetch_next_override.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kt_admit_launch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p]
        L.kt_fetch_reserved.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(S.KtAmounts)]
        _LIB = L
    return _LIB


class ReconcileResult:
    """Rows [0, n) of a reconcile pass; attribute names as in the oracle's result."""

    def __init__(self, n, D):
        self.used, self.calc = S.Amounts(n, D), S.Amounts(n, D)
        m = max(n, 1)
        self.calc_updated = np.zeros(m, np.uint8)
        self.thrl_flag = np.zeros(m, np.uint32)
        self.thrl_has = np.zeros(m, np.uint32)
        self.thrl_pod = np.zeros(m, np.uint8)
        self.error = np.zeros(m, np.uint8)

    def as_struct(self) -> KtStatus:
        p8, p32 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        return KtStatus(self.used.as_struct(), self.calc.as_struct(), self.calc_updated.ctypes.data_as(p8),
                        self.thrl_flag.ctypes.data_as(p32), self.thrl_has.ctypes.data_as(p32),
                        self.thrl_pod.ctypes.data_as(p8), None, self.error.ctypes.data_as(p8))


class Engine:
    """One engine = one GPU's worth of pod rows plus the (replicated) throttle tables."""

    def __init__(self, n_dims, max_labels, pod_capacity, throttle_capacity, namespace_capacity, device=-1,
                 kernel_variant=VARIANT_INDEXED):
        self._h = C.c_void_p()
        self.D = n_dims
        cfg = KtConfig(n_dims, max_labels, pod_capacity, throttle_capacity, namespace_capacity, device, kernel_variant)
        rc = lib().kt_engine_create(C.byref(cfg), C.byref(self._h))
        if rc != KT_OK:
            raise EngineError(rc, lib().kt_last_error(None).decode())

    @classmethod
    def for_snapshot(cls, snap: S.Snapshot, kernel_variant=VARIANT_INDEXED, device=-1, pod_capacity=None):
        e = cls(snap.D, max(snap.L, 1), pod_capacity or max(snap.n_pods, 1), max(snap.n_thr, 1), max(snap.n_ns, 1),
                device, kernel_variant)
        e.load_snapshot(snap)
        return e

    def close(self):
        if self._h:
            lib().kt_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != KT_OK:
            raise EngineError(rc, lib().kt_last_error(self._h).decode())

    # ---- state feed
    def load_snapshot(self, snap: S.Snapshot):
        st = snap.as_struct()
        self._ck(lib().kt_load_snapshot(self._h, C.byref(st)))

    @staticmethod
    def _rows(rows, dtype):
        if rows is None:
            return None, None
        a = np.ascontiguousarray(rows, dtype=dtype)
        return a, a.ctypes.data

    def upsert_namespaces(self, batch: S.Snapshot, rows=None):
        st = batch.as_struct()
        a, p = self._rows(rows, np.int32)
        self._ck(lib().kt_upsert_namespaces(self._h, C.byref(st), p))

    def upsert_pods(self, batch: S.Snapshot, rows=None):
        st = batch.as_struct()
        a, p = self._rows(rows, np.int64)
        self._ck(lib().kt_upsert_pods(self._h, C.byref(st), p))

    def upsert_throttles(self, batch: S.Snapshot, rows=None):
        st = batch.as_struct()
        a, p = self._rows(rows, np.int32)
        self._ck(lib().kt_upsert_throttles(self._h, C.byref(st), p))

    def delete_pods(self, rows):
        a, p = self._rows(rows, np.int64)
        self._ck(lib().kt_delete_pods(self._h, len(a), p))

    def delete_throttles(self, rows):
        a, p = self._rows(rows, np.int32)
        self._ck(lib().kt_delete_throttles(self._h, len(a), p))

    def delete_namespaces(self, rows):
        a, p = self._rows(rows, np.int32)
        self._ck(lib().kt_delete_namespaces(self._h, len(a), p))

    def set_reserved(self, rows, amounts: S.Amounts):
        a, p = self._rows(rows, np.int32)
        st = amounts.as_struct()
        self._ck(lib().kt_set_reserved(self._h, len(a), p, C.byref(st)))

    def set_status(self, rows, used: S.Amounts, calc: S.Amounts, calc_at_nonzero, thrl_flag, thrl_has, thrl_pod,
                   msgs_fp=None):
        a, p = self._rows(rows, np.int32)
        n = len(a)
        keep = [np.ascontiguousarray(calc_at_nonzero, np.uint8), np.ascontiguousarray(thrl_flag, np.uint32),
                np.ascontiguousarray(thrl_has, np.uint32), np.ascontiguousarray(thrl_pod, np.uint8),
                np.ascontiguousarray(msgs_fp if msgs_fp is not None else np.zeros(n), np.uint64)]
        st = KtStatus(used.as_struct(), calc.as_struct(), keep[0].ctypes.data_as(C.POINTER(C.c_uint8)),
                      keep[1].ctypes.data_as(C.POINTER(C.c_uint32)), keep[2].ctypes.data_as(C.POINTER(C.c_uint32)),
                      keep[3].ctypes.data_as(C.POINTER(C.c_uint8)), keep[4].ctypes.data_as(C.POINTER(C.c_uint64)), None)
        self._ck(lib().kt_set_status(self._h, n, p, C.byref(st)))

    # ---- reconcile
    def reconcile_launch(self, now, apply=True, stream=None):
        self._ck(lib().kt_reconcile_launch(self._h, int(now[0]), int(now[1]), RECONCILE_APPLY if apply else 0, stream))

    def reconcile_rows(self, now, rows, apply=True) -> "ReconcileResult":
        """Reconcile of the listed throttle rows only (one workqueue key each, throttle_controller.go:84-133)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self._ck(lib().kt_reconcile_rows_launch(self._h, int(now[0]), int(now[1]), RECONCILE_APPLY if apply else 0,
                                                len(rows), rows.ctypes.data_as(C.c_void_p), None))
        return self.reconcile_fetch()

    def aggregate_launch(self, stream=None):
        self._ck(lib().kt_aggregate_launch(self._h, stream))

    def partial_used_buffer(self):
        ptr, n = C.c_void_p(), C.c_int64()
        self._ck(lib().kt_partial_used_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def use_partial_buffer(self, device_ptr, n_int64):
        self._ck(lib().kt_use_partial_buffer(self._h, device_ptr, n_int64))

    # ---- native RCCL exchange of the partials (kt_comm_*): multi-GPU without a framework in the process
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib().kt_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(rc, lib().kt_last_error(None).decode())
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        self._ck(lib().kt_comm_init(self._h, rank, world, C.create_string_buffer(unique_id, 128)))

    def comm_allreduce_partial(self, stream=None):
        self._ck(lib().kt_comm_allreduce_partial(self._h, stream))

    def comm_destroy(self):
        self._ck(lib().kt_comm_destroy(self._h))

    def reconcile_fetch_used_hi(self, n=None):
        """High 64 bits of the last reconcile's `used` values -> (int64 [n][D], any value beyond int64?)."""
        n = self.throttle_rows() if n is None else n
        hi = np.zeros((max(n, 1), self.D), np.int64)
        anyw = C.c_int32()
        self._ck(lib().kt_reconcile_fetch_used_hi(self._h, n, hi.ctypes.data_as(C.c_void_p), C.byref(anyw)))
        return hi[:n], bool(anyw.value)

    def few_checks_served(self) -> int:
        """kt_check calls that took the few-pod path (kt_kernels_few.hip) so far."""
        return int(lib().kt_counter(self._h, 0))

    def compiles(self) -> int:
        """Selector program compiles + index builds so far (Throttle events that leave the selectors alone do not count)."""
        return int(lib().kt_counter(self._h, 1))

    def set_exchange_world(self, world: int):
        """Ranks whose partials the CALLER sums with its own collective (kt_comm_init declares it by itself)."""
        self._ck(lib().kt_set_exchange_world(self._h, world))

    def affected_pods(self, pod_rows, throttle_rows):
        """kt_affected_pods: [n][m] — 1 where the throttle's selector matches the pod as held now (affectedPods restricted to
        the named pods), 0 where not, 255 for a pod whose PreFilter is an error."""
        pr = np.ascontiguousarray(pod_rows, dtype=np.int64)
        tr = np.ascontiguousarray(throttle_rows, dtype=np.int32)
        out = np.zeros((max(len(pr), 1), max(len(tr), 1)), np.uint8)
        self._ck(lib().kt_affected_pods(self._h, len(pr), pr.ctypes.data, len(tr), tr.ctypes.data, out.ctypes.data))
        return out[:len(pr), :len(tr)]

    def reload_env(self):
        """kt_debug_reload_env: re-read the A/B switches (KT_NO_* environment variables) — they are read once, at creation."""
        self._ck(lib().kt_debug_reload_env(self._h))

    def index_stats(self) -> dict:
        """The compiled selector index (after the first launch): LDS-sized chunks, 64-bit words of term numbers, namespace rows,
        and the words a pod visits on average over the namespace rows in use."""
        ch, words, visits, rows, cvis, iw, slow = (int(lib().kt_counter(self._h, k)) for k in (
            COUNTER_INDEX_CHUNKS, COUNTER_INDEX_WORDS, COUNTER_NS_WORD_VISITS, COUNTER_NS_ROWS, COUNTER_NS_CHUNK_VISITS, COUNTER_INDEX_IMAGE_WORDS,
            COUNTER_SLOW_THROTTLES))
        return {"chunks": ch, "words": words, "image_words": iw, "namespace_rows": rows,
                "word_visits_per_namespace": round(visits / rows, 3) if rows > 0 else None,
                "chunks_per_namespace": round(cvis / rows, 3) if rows > 0 else None, "slow_throttles": slow}

    def packed_words(self) -> int:
        """64-bit words per pod of the packed fold the last full aggregate scan ran with (0: the plain fold)."""
        return int(lib().kt_counter(self._h, COUNTER_PACKED_WORDS))

    def partial_words(self) -> int:
        return self.throttle_rows() * partial_layout(self.D)["stride"]

    def set_wide_sums(self, mode: int):
        """1: always the two-block (limb sums) form of the partial buffer — what every rank of a multi-rank exchange must
        agree on; 0: decided per engine (single rank)."""
        self._ck(lib().kt_set_wide_sums(self._h, mode))

    def pending_partial_words(self):
        """(int64 words, two-block form?) of the aggregate that is pending."""
        n, w = C.c_int64(), C.c_int32()
        self._ck(lib().kt_partial_words(self._h, C.byref(n), C.byref(w)))
        return int(n.value), bool(w.value)

    def finalize_launch(self, now, apply=True, stream=None):
        self._ck(lib().kt_finalize_launch(self._h, int(now[0]), int(now[1]), RECONCILE_APPLY if apply else 0, stream))

    def throttle_rows(self) -> int:
        n = C.c_int32()
        self._ck(lib().kt_throttle_rows(self._h, C.byref(n)))
        return n.value

    def reconcile_fetch(self, n=None) -> ReconcileResult:
        n = self.throttle_rows() if n is None else n
        r = ReconcileResult(n, self.D)
        st = r.as_struct()
        self._ck(lib().kt_reconcile_fetch(self._h, n, C.byref(st)))
        return r

    def next_override(self, n=None):
        """NextOverrideHappensIn of the last reconcile: (instant seconds, nanoseconds, has) per throttle row."""
        n = self.throttle_rows() if n is None else n
        sec, nsec, has = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.uint8)
        self._ck(lib().kt_reconcile_fetch_next_override(self._h, n, sec.ctypes.data, nsec.ctypes.data, has.ctypes.data))
        return sec[:n], nsec[:n], has[:n]

    def reconcile(self, now, apply=True) -> ReconcileResult:
        self.reconcile_launch(now, apply)
        return self.reconcile_fetch()

    # ---- check
    def check_launch(self, n, rows=None, on_equal=False, want_status=False, stream=None):
        a, p = self._rows(rows, np.int64)
        self._ck(lib().kt_check_launch(self._h, n, p, int(on_equal), CHECK_STATUS_MATRIX if want_status else 0, stream))

    def sweep_launch(self, now, apply=True, on_equal=False, stream=None):
        """kt_sweep_launch: PreFilter sweep of every pod row against the stored status + reconcile of every throttle, one pass
        over the pod tables; read the results with check_fetch(pod rows) / reconcile_fetch()."""
        self._ck(lib().kt_sweep_launch(self._h, int(now[0]), int(now[1]), RECONCILE_APPLY if apply else 0, int(on_equal), stream))

    def check_fetch(self, n, want_status=False):
        T = self.throttle_rows()
        summary = np.zeros(max(n, 1), np.uint64)
        status = np.zeros((max(n, 1), max(T, 1)), np.uint8) if want_status else None
        self._ck(lib().kt_check_fetch(self._h, n, summary.ctypes.data, None if status is None else status.ctypes.data))
        return (None if status is None else status[:n, :T]), summary[:n]

    def check(self, rows=None, n=None, on_equal=False, want_status=True):
        n = len(rows) if rows is not None else n
        self.check_launch(n, rows, on_equal, want_status)
        return self.check_fetch(n, want_status)

    def check_atomic(self, rows=None, n=None, on_equal=False, want_status=True):
        """kt_check: launch + fetch under one engine lock (safe next to other threads using the engine)."""
        a, p = self._rows(rows, np.int64)
        n = len(a) if a is not None else n
        T = self.throttle_rows()
        summary = np.zeros(max(n, 1), np.uint64)
        status = np.zeros((max(n, 1), max(T, 1)), np.uint8) if want_status else None
        self._ck(lib().kt_check(self._h, n, p, int(on_equal), summary.ctypes.data,
                                None if status is None else status.ctypes.data))
        return (None if status is None else status[:n, :T]), summary[:n]

    def checker(self, n: int, on_equal=False):
        """Pre-bound kt_check(n pods, summaries only) for latency-sensitive callers: (rows, summary, call) — fill rows[:],
        call(), read summary[:]; no per-call allocation or argument marshalling beyond the foreign call itself."""
        rows = np.zeros(n, np.int64)
        summary = np.zeros(n, np.uint64)
        fn, h = lib().kt_check, self._h
        rp, sp, eq = C.c_void_p(rows.ctypes.data), C.c_void_p(summary.ctypes.data), int(on_equal)

        def call():
            rc = fn(h, n, rp, eq, sp, None)
            if rc != KT_OK:
                self._ck(rc)
        return rows, summary, call

    # ---- sequential admission with reservation (N1): results are read like a check's
    def admit(self, rows=None, n=None, on_equal=False, commit=False, want_status=True):
        a, p = self._rows(rows, np.int64)
        n = len(a) if a is not None else n
        self._ck(lib().kt_admit_launch(self._h, n, p, int(on_equal), ADMIT_COMMIT if commit else 0, None))
        return self.check_fetch(n, want_status)

    def fetch_reserved(self, rows=None) -> S.Amounts:
        rows = np.arange(self.throttle_rows(), dtype=np.int32) if rows is None else rows
        a, p = self._rows(rows, np.int32)
        out = S.Amounts(len(a), self.D)
        st = out.as_struct()
        self._ck(lib().kt_fetch_reserved(self._h, len(a), p, C.byref(st)))
        return out

    def check_device_summary(self) -> int:
        ptr = C.c_void_p()
        self._ck(lib().kt_check_device_summary(self._h, C.byref(ptr)))
        return ptr.value

    def fetch_pod_requests(self, rows=None, n=None):
        a, p = self._rows(rows, np.int64)
        n = len(a) if a is not None else n
        v = np.zeros((max(n, 1), self.D), np.int64)
        present = np.zeros(max(n, 1), np.uint32)
        self._ck(lib().kt_fetch_pod_requests(self._h, n, p, v.ctypes.data, present.ctypes.data))
        return v[:n], present[:n]

    # ---- measurement
    def timing_enable(self, on=True):
        self._ck(lib().kt_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._ck(lib().kt_timing_reset(self._h))

    def timing_read(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._ck(lib().kt_timing_read(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def synchronize(self, stream=None):
        self._ck(lib().kt_synchronize(self._h, stream))

    def kernel_name(self, kernel) -> str:
        return lib().kt_kernel_name(self._h, kernel).decode()
