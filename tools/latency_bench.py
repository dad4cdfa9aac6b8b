#!/usr/bin/env python
"""What the scheduler would feel: latencies of the single-object calls of the drop-in boundary (host wall clock, through
the C-ABI, engine already holding the snapshot).

  check1        kt_check(n=1): PreFilter of ONE pod (plugin.go:148-215) — H2D of the row, kernels, D2H of the summary
  check1_busy   the same while another thread runs kt_reconcile_launch + kt_reconcile_fetch in a loop
  native        check1 / check1_busy once more from native threads (tools/native/busy_latency.c): the Python loops above
                share the interpreter lock between the timing thread and the reconciling thread
  upsert_pod1   kt_upsert_pods(1): a pod informer event — the call alone (packs the event into a pinned slot, enqueues
                kt_ingest_pods + kt_translate_pods + kt_patch_scan_views, records an event; no stream synchronisation)
  upsert_pod64_per_pod  a coalesced batch of 64 events through the same call, per pod
  upsert1_then_check1   one event + kt_check(n=1) of that pod: the check waits for the ingest (settle_ingest);
                ..._unfused: with KT_NO_FEED_FUSION=1 (kt_ingest_pods + kt_translate_pods + copy + kt_patch_scan_views
                instead of the single kt_feed_small launch)
  upsert_pod1_blocking  the form of rounds 1-3 (KT_SYNC_INGEST=1: device staging copy + stream synchronisation)
  delete_pod1   kt_delete_pods(1)
  sweep         a full reconcile + PreFilter sweep in the steady state, and the first one after ONE pod event (which rebuilds
                the scan lists, the scan-ordered record copies and the request-sum proof of the overflow guard)
  throttle_event  the first kt_check after ONE kt_upsert_throttles that leaves the row's selector as stored (threshold edit,
                status update): the throttle tables go up again, nothing is compiled
  recompile     the first kt_check after ONE kt_upsert_throttles that changes a selector: selector program recompile +
                index rebuild + upload + re-translation of every pod's labels (also what any namespace event costs)
(bench.py's cpu_baseline leg adds the CPU restatement's PreFilter of one pod on one core as the yardstick.)

    python tools/latency_bench.py --config 2        (prints one JSON object)
bench.py embeds the same object as "latency" (N = 1 runs)."""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pct(xs):
    a = np.sort(np.asarray(xs)) * 1e6
    return {"p50_us": round(float(a[len(a) // 2]), 1), "p99_us": round(float(a[min(len(a) - 1, int(len(a) * 0.99))]), 1),
            "mean_us": round(float(a.mean()), 1), "n": int(len(a))}


def _native_calls(eng, rows, now):
    """The same two measurements from native threads (tools/native/busy_latency.c): no interpreter lock between the
    timing thread and the reconciling thread — what a cgo caller sees.  None when the shim cannot be built."""
    import ctypes as C
    import subprocess
    here = os.path.join(ROOT, "tools", "native")
    so = os.path.join(here, "_busy_latency.so")
    src = os.path.join(here, "busy_latency.c")
    csrc = os.path.join(ROOT, "kube_throttler_amd", "csrc")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), src, "-L" + csrc, "-lkt_engine", "-lpthread",
                                   "-Wl,-rpath," + csrc, "-o", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(so)
    except Exception:
        return None
    fn = lib.kt_native_check_latency
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    fn.restype = C.c_int64
    out = {}
    for key, busy, n in (("check1", 0, len(rows)), ("check1_busy", 1, max(1000, len(rows) // 2))):
        r = np.ascontiguousarray(rows[:n], dtype=np.int64)
        us = np.zeros(n, dtype=np.float64)
        rc = int(fn(eng._h, int(now[0]), n, r.ctypes.data_as(C.c_void_p), us.ctypes.data_as(C.c_void_p), busy))
        if rc < 0:
            return {"error": rc}
        out[key] = _pct(us * 1e-6)
        if busy:
            out[key]["reconciles_meanwhile"] = rc
    return out


def measure(eng, snap, n_check=10000, n_upsert=300, now=(1767225600, 0)):
    from kube_throttler_amd import snapshot as S
    P = snap.n_pods
    rng = np.random.default_rng(7)
    rows = rng.integers(0, P, size=n_check).astype(np.int64)
    out = {}
    eng.reconcile(now, apply=True)
    eng.check_atomic(rows=rows[:1], want_status=False)  # warm: CheckRecs built, buffers sized

    row1, _sum1, call1 = eng.checker(1)  # pre-bound kt_check(n=1): the foreign call is all that is timed

    def sweep(rs):
        ts = []
        for r in rs:
            row1[0] = r
            t0 = time.perf_counter()
            call1()
            ts.append(time.perf_counter() - t0)
        return ts
    out["check1"] = _pct(sweep(rows))
    # ---- the same next to a reconcile loop on another thread
    stop = threading.Event()
    n_rec = [0]

    def reconciler():
        while not stop.is_set():
            eng.reconcile(now, apply=True)
            n_rec[0] += 1
    th = threading.Thread(target=reconciler)
    th.start()
    try:
        busy = sweep(rows[:max(1000, n_check // 4)])
    finally:
        stop.set()
        th.join()
    out["check1_busy"] = dict(_pct(busy), reconciles_meanwhile=n_rec[0])
    nat = _native_calls(eng, rows, now)
    if nat is not None:
        out["native"] = nat
    out["few_path_checks"] = eng.few_checks_served()  # kt_check calls served without copy / stream sync (kt_check_few)
    # ---- pod informer events (throttle_controller.go:400-536: one handler call per event).  Only the foreign call is
    #      timed (batch, struct and row array are built before); the call returns once its kernels are enqueued, the next
    #      entry point that is not a pod feed call waits for them (settle_ingest) — upsert1_then_check1 is that sum.
    from kube_throttler_amd import engine as E
    lib = E.lib()

    def feed(batch, reps, then_check=False, per_pod=True):
        ts = []
        for _ in range(reps):
            rs = np.unique(rng.integers(0, P, size=batch).astype(np.int64))
            one = snap.pod_batch(rs)
            st = one.as_struct()
            rp = rs.ctypes.data_as(C.c_void_p)
            row1[0] = rs[0]
            t0 = time.perf_counter()
            rc = lib.kt_upsert_pods(eng._h, C.byref(st), rp)
            if then_check:
                call1()
            t1 = time.perf_counter()
            assert rc == 0
            ts.append((t1 - t0) / (len(rs) if per_pod else 1))
        eng.synchronize()
        return _pct(ts)
    feed(1, 20)
    out["upsert_pod1"] = feed(1, n_upsert)                           # one event, the call alone
    out["upsert_pod64_per_pod"] = feed(64, max(20, n_upsert // 4))   # a coalesced batch of 64 events, per pod
    out["upsert1_then_check1"] = feed(1, n_upsert, then_check=True)  # event + PreFilter of that pod (waits for the ingest)
    # (the engine reads its A/B switches once: reload_env after every flip)
    os.environ["KT_NO_FEED_FUSION"] = "1"                            # three launches + a copy instead of kt_feed_small
    eng.reload_env()
    try:
        out["upsert1_then_check1_unfused"] = feed(1, max(50, n_upsert // 2), then_check=True)
    finally:
        del os.environ["KT_NO_FEED_FUSION"]
        eng.reload_env()
    os.environ["KT_SYNC_INGEST"] = "1"                               # the blocking form of rounds 1-3, for comparison
    eng.reload_env()
    try:
        out["upsert_pod1_blocking"] = feed(1, max(50, n_upsert // 2))
    finally:
        del os.environ["KT_SYNC_INGEST"]
        eng.reload_env()
    ts = []
    for r in rng.integers(0, P, size=max(50, n_upsert // 2)):
        rs = np.array([int(r)], dtype=np.int64)
        t0 = time.perf_counter()
        lib.kt_delete_pods(eng._h, 1, rs.ctypes.data_as(C.c_void_p))
        ts.append(time.perf_counter() - t0)
        eng.upsert_pods(snap.pod_batch(rs), rows=rs)                 # the pod comes back: the state stays configs' own
    out["delete_pod1"] = _pct(ts)
    eng.synchronize()
    # ---- a full sweep (reconcile + PreFilter of every pod) right after ONE pod event, against the steady sweep: the event
    #      voids the scan lists / scan-ordered record copies / request-sum proof, the next sweep rebuilds them
    def full_sweep():
        t0 = time.perf_counter()
        eng.reconcile_launch(now, True)
        eng.check_launch(P, None, False, False)
        eng.synchronize()
        return time.perf_counter() - t0
    full_sweep()
    steady = [full_sweep() for _ in range(20)]
    after = []
    for r in rng.integers(0, P, size=20):
        one = snap.pod_batch(np.array([int(r)], dtype=np.int64))
        eng.upsert_pods(one, rows=np.array([int(r)], dtype=np.int64))
        eng.synchronize()  # the feed call no longer waits for its kernels: what is compared is the SWEEP (rebuilt views or not)
        after.append(full_sweep())
    out["sweep"] = {"steady_ms": round(float(np.median(steady)) * 1e3, 3), "after_pod_event_ms": round(float(np.median(after)) * 1e3, 3), "n": 20}
    # ---- one Throttle event that leaves every selector as it is (a threshold edit, the controller's own status update
    #      coming back): the throttle tables are uploaded again, program and index stand
    c0 = eng.compiles()
    ts = []
    for k in range(5):
        t = int(rng.integers(0, snap.n_thr))
        one = snap.throttle_batch(np.array([t], dtype=np.int32))
        eng.upsert_throttles(one, rows=np.array([t], dtype=np.int32))
        t0 = time.perf_counter()
        eng.check_atomic(rows=rows[:1], want_status=False)
        ts.append(time.perf_counter() - t0)
    out["throttle_event"] = {"mean_ms": round(float(np.mean(ts)) * 1e3, 3), "min_ms": round(float(np.min(ts)) * 1e3, 3), "n": len(ts),
                             "compiles": eng.compiles() - c0}
    # ---- one Throttle event that changes what a selector selects (here: the throttle stops / starts being this
    #      scheduler's): the next call recompiles the program, rebuilds the index and re-translates the pods — what a
    #      selector edit or ANY namespace event costs
    c0 = eng.compiles()
    ts = []
    t = int(rng.integers(0, snap.n_thr))
    for k in range(6):  # an even number of flips: the snapshot ends as it began
        snap.thr_flags[t] ^= S.THR_RESPONSIBLE
        one = snap.throttle_batch(np.array([t], dtype=np.int32))
        eng.upsert_throttles(one, rows=np.array([t], dtype=np.int32))
        t0 = time.perf_counter()
        eng.check_atomic(rows=rows[:1], want_status=False)
        ts.append(time.perf_counter() - t0)
    out["recompile"] = {"mean_ms": round(float(np.mean(ts)) * 1e3, 2), "min_ms": round(float(np.min(ts)) * 1e3, 2), "n": len(ts),
                        "throttles": int(snap.n_thr), "pods": int(P), "compiles": eng.compiles() - c0}
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--pods", type=int, default=0)
    args = ap.parse_args()
    from kube_throttler_amd import engine as E, workload as W
    cfg = W.preset(args.config)
    if args.config == 4 and not args.pods:
        cfg = cfg.shard(0, 8)
    if args.pods:
        cfg.n_pods_total = cfg.n_pods = args.pods
    snap = W.generate(cfg)
    eng = E.Engine.for_snapshot(snap)
    try:
        res = measure(eng, snap, now=(cfg.now_s, 0))
    finally:
        eng.close()
    print(json.dumps({"workload": f"configs[{args.config}]", "pods": int(snap.n_pods), "throttles": int(snap.n_thr), "latency": res}))


if __name__ == "__main__":
    main()
