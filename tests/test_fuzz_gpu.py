"""Randomised shapes through the whole path (GPU): every case draws its own cluster shape — dimensions, labels per pod, key /
value universe, namespaces, terms per throttle, requirements per term, rich operators, invalid selectors, missing namespaces —
AND the switches that select kernel instantiations (index chunk budget, rank windows of the aggregate, incremental engine), then
runs `run_full_parity` (reconcile -> stored status -> check matrix + lean sweep + fused sweep, both isThrottledOnEqual values)
against the oracle.  The fixed-shape tests of test_engine_gpu.py pin the forms one by one; this file crosses them.

KT_FUZZ_CASES (default 24, ~1 minute) sizes the run, KT_FUZZ_SEED moves it; a failing case prints the line that reproduces it:
    KT_FUZZ_ONLY=<case seed> python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q
"""
import os

import numpy as np
import pytest

from kube_throttler_amd import engine as E
from kube_throttler_amd import snapshot as S
from kube_throttler_amd import workload as W
from test_engine_gpu import run_full_parity

pytestmark = pytest.mark.gpu

# the A/B switches of the engine (kt_engine_impl.h: EnvSwitch, read at engine creation): every one selects a path that stays in the
# library — the plain fold, scans that gather through the row lists, unfused reductions, feeds without the few-row kernels ...
PATH_SWITCHES = ("KT_FEED_NO_STAGE", "KT_FORCE_NS_ORDER", "KT_NO_FEED_FEW", "KT_NO_FEED_FUSION", "KT_NO_FUSED", "KT_NO_NS_ORDER", "KT_NO_PACK",
                 "KT_NO_SCAN_VIEW", "KT_NO_SWEEP", "KT_NO_VERDICT_IMAGES", "KT_NO_WG_RANGES", "KT_SYNC_INGEST", "KT_NO_VIEW_PATCH",
                 "KT_CHECK_ONE_PER_CU", "KT_INGEST_EVENT_WAIT", "KT_INGEST_TRUST_FENCE")
SWITCHES = ("KT_CHUNK_BUDGET", "KT_AGG_SMALL_WINDOW", "KT_CUT_PLAN") + PATH_SWITCHES


def draw_case(seed):
    """(workload config kwargs, environment switches, engine variant, post-processing of the snapshot) of one case."""
    r = np.random.default_rng(seed)
    D = int(r.choice([1, 2, 3, 4, 5, 8, 8, 9, 12, 16]))
    L = int(r.choice([1, 2, 4, 6, 8, 8, 12, 16, 24, 30]))
    t_hi = int(r.integers(1, 6))
    q_hi = int(r.integers(1, 6))
    K = int(min(64, max(L, q_hi, int(r.integers(L, 2 * L + 4)))))  # (a term names distinct keys: the generator needs K >= requirements)
    V = int(r.integers(2, 9))
    n_thr = int(r.choice([8, 24, 64, 96, 160, 300, 420]))
    kw = dict(seed=int(r.integers(1, 1 << 30)), n_pods=int(r.integers(300, 5000)), n_thr=n_thr, n_cluster=int(r.integers(0, n_thr + 1)), D=D,
              n_ns=int(r.integers(1, 40)), K=K, V=V, L=L, terms=(1, t_hi), reqs=(int(r.integers(0, 2)), q_hi), rich_ops=int(r.integers(0, 2)),
              overrides=int(r.integers(0, 2)))
    if r.random() < .06:  # throttles with more than 64 terms: runs of term numbers across words (the match-by-match forms)
        kw.update(n_thr=int(r.choice([8, 24, 48])), terms=(1, int(r.integers(66, 140))))
        kw["n_cluster"] = int(r.integers(0, kw["n_thr"] + 1))
    if r.random() < .1:
        kw["n_pods"] = int(r.integers(5000, 30000))
    if r.random() < .25:
        kw.update(n_invalid_pod_sel=int(r.integers(0, 3)), n_invalid_ns_sel=int(r.integers(0, 3)), n_missing_ns=int(r.integers(0, 2)))
    env = {}
    if r.random() < .5:
        env["KT_CHUNK_BUDGET"] = str(int(r.choice([5000, 9000, 16000, 30000])))
    if r.random() < .3:
        env["KT_AGG_SMALL_WINDOW"] = "1"
    if r.random() < .2:
        env["KT_CUT_PLAN"] = "grouped"
    variant = E.VARIANT_INDEXED | (E.VARIANT_INCREMENTAL if r.random() < .2 else 0)
    post = str(r.choice(["none", "none", "none", "negative", "large", "zero"]))
    r2 = np.random.default_rng(seed ^ 0x5317C4)  # (its own stream: the shapes above stay what they were before these were drawn)
    if r2.random() < .4:
        for k in r2.choice(PATH_SWITCHES, size=int(r2.integers(1, 4)), replace=False):
            env[str(k)] = "1"
    if r2.random() < .08 and not (variant & E.VARIANT_INCREMENTAL):
        variant = E.VARIANT_DENSE  # the dense cross-check kernels (no index)
    if r2.random() < .02:
        kw["n_pods"] = int(r2.integers(60000, 200000))  # hundreds of workgroups: planned ranges, every slab in use
    return kw, env, variant, post


def shape_snapshot(snap, post, seed):
    """A few request populations the generator does not draw: a negative request (the plain fold for good), large values with odd
    low bits (wider packed fields, more packed words), pods that request nothing."""
    r = np.random.default_rng(seed ^ 0x5EED)
    nc = int(snap.pod_ctr_off[snap.n_pods])
    if nc == 0 or post == "none":
        return
    if post == "negative":
        snap.ctr_req[int(r.integers(0, nc)), int(r.integers(0, snap.D))] = -int(r.integers(1, 1000))
        snap.ctr_present[:nc] |= (1 << snap.D) - 1
    elif post == "large":
        d = int(r.integers(0, snap.D))
        snap.ctr_req[:nc, d] = np.where(snap.ctr_req[:nc, d] != 0, ((snap.ctr_req[:nc, d] & 0xFFFFF) << int(r.integers(4, 16))) | 1, 0)
    elif post == "zero":
        snap.ctr_req[:nc:3, :] = 0


def case_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930"))
    return [base + 7919 * i for i in range(int(os.environ.get("KT_FUZZ_CASES", "24")))]


@pytest.mark.parametrize("seed", case_seeds())
def test_random_shape_and_switches(seed, oracle_mod, monkeypatch):
    kw, env, variant, post = draw_case(seed)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    snap = W.generate(W.small(**kw))
    shape_snapshot(snap, post, seed)
    try:
        run_full_parity(snap, oracle_mod, variant, on_equals=(False, True) if seed % 2 else (True,))
    except Exception:
        print(f"\nfuzz case {seed}: KT_FUZZ_ONLY={seed}  shape={kw}  env={env}  variant={variant}  requests={post}")
        raise


def event_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930")) + 13
    return [base + 104729 * i for i in range(int(os.environ.get("KT_FUZZ_EVENT_CASES", "8")))]


@pytest.mark.parametrize("seed", event_seeds())
def test_random_event_sequences(seed, oracle_mod, monkeypatch):
    """Pod events on a random shape: bursts of adds / rewrites / deletes of random size (single pods: the lists and scan views are
    patched in place; larger batches: rebuilt; the same row several times in one batch: the last entry wins) between sweeps of an
    engine that may be incremental, with a Throttle event now and then; after every burst reconcile + lean check equal the oracle
    on the pods currently held."""
    from test_engine_gpu import NOW, _permute_pods, _rows_of, _with_pods, assert_reconcile_equal, responsible_rows
    kw, env, variant, post = draw_case(seed)
    kw["n_pods"] = min(kw["n_pods"], 3000)
    kw.pop("n_missing_ns", None)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    base = W.generate(W.small(**kw))
    shape_snapshot(base, post if post != "negative" else "none", seed)
    r = np.random.default_rng(seed ^ 0xE7E27)
    n_src = base.n_pods
    P = max(64, int(n_src * 0.8))
    state = np.full(P, -1, dtype=np.int64)
    n0 = P // 2
    state[:n0] = r.integers(0, n_src, n0)
    eng = E.Engine(base.D, max(base.L, 1), P, max(base.n_thr, 1), max(base.n_ns, 1), -1, variant)
    try:
        eng.upsert_namespaces(base)
        eng.upsert_throttles(base)
        eng.upsert_pods(_permute_pods(base, state[:n0]), rows=np.arange(n0))

        def sweep(step):
            used = np.nonzero(state >= 0)[0]
            n = int(used.max()) + 1 if len(used) else 1
            snap = _with_pods(base, state[:n])
            o = oracle_mod.Oracle(snap)
            rows = responsible_rows(snap)
            want = o.reconcile(NOW, rows=rows)
            got = _rows_of(eng.reconcile(NOW, apply=True), rows, snap.D)
            try:
                assert_reconcile_equal(got, want, len(rows))
                snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows)
                _, sm_w = o.check(on_equal=bool(step & 1), want_status=False, nthreads=4)
                _, sm_g = eng.check(n=n, on_equal=bool(step & 1), want_status=False)
                np.testing.assert_array_equal(sm_g, sm_w, err_msg=f"summary words after burst {step}")
            except Exception:
                print(f"\nevent fuzz case {seed} (burst {step}): KT_FUZZ_ONLY={seed}  shape={kw}  env={env}  variant={variant}  requests={post}")
                raise
            base.thr_used, base.thr_calc = snap.thr_used, snap.thr_calc
            base.thr_flags, base.thr_thrl_flag, base.thr_thrl_has = snap.thr_flags, snap.thr_thrl_flag, snap.thr_thrl_has

        sweep(0)
        for step in range(1, 9):
            kind = r.random()
            size = int(r.choice([1, 1, 2, 7, 40, 300]))
            if kind < .6:      # upserts (adds and rewrites), rows may repeat inside the batch: the last entry wins
                rows = r.integers(0, P, size)
                src = r.integers(0, n_src, size)
                eng.upsert_pods(_permute_pods(base, src), rows=rows.astype(np.int64))
                for i in range(size):
                    state[rows[i]] = src[i]
            elif kind < .85:   # deletes
                rows = np.unique(r.integers(0, P, size)).astype(np.int64)
                eng.delete_pods(rows)
                state[rows] = -1
            else:              # a Throttle event: the tables go up again (selectors unchanged), or a responsibility flip (recompile)
                rr = responsible_rows(base)
                if len(rr) > 1 and r.random() < .5:
                    t = int(rr[int(r.integers(0, len(rr)))])
                    base.thr_flags[t] &= 0xFFFFFFFF ^ S.THR_RESPONSIBLE
                    eng.upsert_throttles(base.throttle_batch([t]), rows=np.array([t], dtype=np.int32))
                else:
                    eng.upsert_throttles(base)
            sweep(step)
    finally:
        eng.close()


def entry_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930")) + 29
    return [base + 15485863 * i for i in range(int(os.environ.get("KT_FUZZ_ENTRY_CASES", "8")))]


@pytest.mark.parametrize("seed", entry_seeds())
def test_random_shapes_through_the_other_entry_points(seed, oracle_mod, monkeypatch):
    """The entry points beside the two sweeps, on a random shape: reconciles of single workqueue keys (the others keep their stored
    status), single PreFilter calls (kt_check with n <= 8: the few-pod kernels, their own scan), the status row of listed pods,
    `affectedPods` of listed (pod, throttle) pairs against the status matrix, and a pending queue admitted in order with reservation
    (dry run, then committed) — each against the oracle."""
    from test_engine_gpu import NOW, _rows_of, _stored_status, assert_reconcile_equal, responsible_rows
    kw, env, variant, post = draw_case(seed)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    snap = W.generate(W.small(**kw))
    shape_snapshot(snap, post, seed)
    r = np.random.default_rng(seed ^ 0xA11)
    T, P = snap.n_thr, snap.n_pods
    _stored_status(snap, oracle_mod)        # what UpdateStatus persisted before this engine came up
    # (the admission queue wants head-room: thresholds doubled AFTER the status was stored is what a spec edit looks like)
    o = oracle_mod.Oracle(snap)
    eng = E.Engine.for_snapshot(snap, variant & ~E.VARIANT_INCREMENTAL)
    try:
        resp = responsible_rows(snap)
        try:
            # ---- single keys
            if len(resp):
                keys = np.sort(r.choice(resp, size=max(1, len(resp) // int(r.integers(2, 6))), replace=False))
                want = o.reconcile(NOW, rows=keys)
                got_all = eng.reconcile_rows(NOW, keys.astype(np.int32), apply=True)
                assert_reconcile_equal(_rows_of(got_all, keys, snap.D), want, len(keys))
                snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=keys)
            # ---- PreFilter calls of 1..8 pods, and the status rows of a few more
            on_equal = bool(seed & 1)
            for n in (1, 1, 2, 5, 8):
                rows = r.integers(0, P, n).astype(np.int64)
                st_w, sm_w = o.check(rows, on_equal=on_equal)
                _, sm_g = eng.check(rows, on_equal=on_equal, want_status=False)
                np.testing.assert_array_equal(sm_g, sm_w, err_msg=f"kt_check of {n} pods")
            rows = np.unique(r.integers(0, P, 40)).astype(np.int64)
            st_w, sm_w = o.check(rows, on_equal=on_equal)
            st_g, sm_g = eng.check(rows, on_equal=on_equal, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            np.testing.assert_array_equal(sm_g, sm_w)
            # ---- affectedPods of the listed pairs: exactly the cells of the matrix that are not "not affected" (Error rows: 255)
            trs = np.unique(r.integers(0, T, min(T, 12))).astype(np.int32)
            aff = eng.affected_pods(rows, trs)
            err_rows = (S.summary_fields(sm_w)[0] == S.VERDICT_ERROR)
            want_aff = (st_w[:, trs] != S.NOT_AFFECTED).astype(np.uint8)
            want_aff[err_rows] = 255
            valid_thr = (snap.thr_flags[trs] & S.THR_VALID) != 0
            np.testing.assert_array_equal(aff[:, valid_thr], want_aff[:, valid_thr], err_msg="kt_affected_pods")
            # ---- a pending queue admitted in order, reserving on the way
            fl = snap.pod_flags[:P]
            pending = np.nonzero(((fl & S.POD_VALID) != 0) & ((fl & S.POD_SCHEDULED) == 0))[0]
            if len(pending):
                queue = r.permutation(pending)[:int(r.integers(1, 400))].astype(np.int64)
                st_w, sm_w, res_w = o.admit(queue, on_equal=on_equal)
                st_d, sm_d = eng.admit(queue, on_equal=on_equal, commit=False)
                np.testing.assert_array_equal(st_d, st_w)
                np.testing.assert_array_equal(sm_d, sm_w)
                st_g, sm_g = eng.admit(queue, on_equal=on_equal, commit=True)
                np.testing.assert_array_equal(sm_g, sm_w)
                res_g = eng.fetch_reserved()
                for f in ("v", "present", "count", "has_count"):
                    np.testing.assert_array_equal(getattr(res_g, f)[resp], getattr(res_w, f)[resp], err_msg=f"reserved.{f}")
        except Exception:
            print(f"\nentry-point fuzz case {seed}: KT_FUZZ_ONLY={seed}  shape={kw}  env={env}  requests={post}")
            raise
    finally:
        eng.close()


def conc_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930")) + 57
    return [base + 32452843 * i for i in range(int(os.environ.get("KT_FUZZ_CONC_CASES", "3")))]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", conc_seeds())
def test_concurrent_events_prefilters_and_reconciles(seed, oracle_mod, monkeypatch):
    """One engine, four kinds of callers at once (the scheduler's PreFilter goroutines, the informers' pod handlers, the two
    controllers' workers): a thread feeds a pre-drawn sequence of pod upserts / deletes (single rows and small batches), two threads
    issue PreFilter calls of 1-8 pods without pause (the few-pod path under the shared lock) and whole sweeps, one thread
    reconciles (apply) in a loop, one reconciles single keys.  No call may fail or hang; afterwards the engine holds exactly the
    pods of the sequence's end state: reconcile + status matrix equal the oracle on them."""
    import threading
    from test_engine_gpu import NOW, _permute_pods, _rows_of, _with_pods, assert_reconcile_equal, responsible_rows
    kw, env, variant, post = draw_case(seed)
    kw["n_pods"] = min(kw["n_pods"], 2500)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    base = W.generate(W.small(**kw))
    shape_snapshot(base, post if post != "negative" else "none", seed)
    r = np.random.default_rng(seed ^ 0xC0C0)
    n_src = base.n_pods
    P = max(64, int(n_src * 0.8))
    state = np.full(P, -1, dtype=np.int64)
    n0 = P // 2
    state[:n0] = r.integers(0, n_src, n0)
    # the event sequence, drawn up front (the feeder thread only replays it)
    events = []
    for _ in range(int(os.environ.get("KT_FUZZ_CONC_EVENTS", "150"))):
        size = int(r.choice([1, 1, 1, 2, 6, 30]))
        if r.random() < .7:
            events.append(("up", r.integers(0, P, size).astype(np.int64), r.integers(0, n_src, size)))
        else:
            events.append(("del", np.unique(r.integers(0, P, size)).astype(np.int64), None))
    eng = E.Engine(base.D, max(base.L, 1), P, max(base.n_thr, 1), max(base.n_ns, 1), -1, variant)
    errors, stop = [], threading.Event()
    hung = False

    def guarded(fn):
        def run(*a):
            try:
                fn(*a)
            except Exception as ex:  # noqa: BLE001 — collected, re-raised on the main thread
                errors.append(ex)
                stop.set()
        return run

    @guarded
    def feeder():
        for kind, rows, src in events:
            if stop.is_set():
                return
            if kind == "up":
                eng.upsert_pods(_permute_pods(base, src), rows=rows)
            else:
                eng.delete_pods(rows)

    @guarded
    def prefilter(k):
        rr = np.random.default_rng(seed + k)
        while not stop.is_set():
            n = int(rr.choice([1, 1, 3, 8]))
            eng.check_atomic(rows=rr.integers(0, P, n).astype(np.int64), on_equal=bool(k & 1), want_status=False)
            if rr.random() < .05:
                eng.check_atomic(n=P, on_equal=False, want_status=False)

    @guarded
    def reconciler():
        while not stop.is_set():
            eng.reconcile(NOW, apply=True)

    @guarded
    def key_worker():
        rr = np.random.default_rng(seed + 99)
        rows = responsible_rows(base)
        while not stop.is_set() and len(rows):
            eng.reconcile_rows(NOW, rr.choice(rows, size=min(len(rows), 3), replace=False).astype(np.int32), apply=True)

    try:
        eng.upsert_namespaces(base)
        eng.upsert_throttles(base)
        eng.upsert_pods(_permute_pods(base, state[:n0]), rows=np.arange(n0))
        eng.reconcile(NOW, apply=True)
        # (both sides start from the same stored status: whether a threshold counts as replaced depends on what was stored)
        snap0 = _with_pods(base, state[:n0])
        rows0 = responsible_rows(snap0)
        w0 = oracle_mod.Oracle(snap0).reconcile(NOW, rows=rows0)
        snap0.apply_status(w0.used, w0.calc, w0.calc_updated, w0.thrl_flag, w0.thrl_has, w0.thrl_pod, w0.error, rows=rows0)
        base.thr_used, base.thr_calc = snap0.thr_used, snap0.thr_calc
        base.thr_flags, base.thr_thrl_flag, base.thr_thrl_has = snap0.thr_flags, snap0.thr_thrl_flag, snap0.thr_thrl_has
        workers = [threading.Thread(target=prefilter, args=(k,), daemon=True) for k in range(2)]
        workers += [threading.Thread(target=reconciler, daemon=True), threading.Thread(target=key_worker, daemon=True)]
        fd = threading.Thread(target=feeder, daemon=True)
        for th in workers:
            th.start()
        fd.start()
        fd.join(timeout=240)
        stop.set()
        for th in workers:
            th.join(timeout=60)
        hung = fd.is_alive() or any(th.is_alive() for th in workers)
        try:
            assert not hung, "a caller did not return"
            if errors:
                raise errors[0]
            for kind, rows, src in events:  # the end state of the sequence (inside a batch the last entry of a row wins)
                if kind == "up":
                    for i in range(len(rows)):
                        state[rows[i]] = src[i]
                else:
                    state[rows] = -1
            used = np.nonzero(state >= 0)[0]
            n = int(used.max()) + 1 if len(used) else 1
            snap = _with_pods(base, state[:n])
            o = oracle_mod.Oracle(snap)
            rows_t = responsible_rows(snap)
            want = o.reconcile(NOW, rows=rows_t)
            got = _rows_of(eng.reconcile(NOW, apply=True), rows_t, snap.D)
            # (calc_updated depends on what the racing reconciles stored before: everything else is a function of the end state)
            got.calc_updated[:] = want.calc_updated[:len(got.calc_updated)]
            assert_reconcile_equal(got, want, len(rows_t))
            snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows_t)
            st_w, sm_w = o.check(on_equal=False, nthreads=4)
            st_g, sm_g = eng.check(n=n, on_equal=False, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            np.testing.assert_array_equal(sm_g, sm_w)
        except Exception:
            print(f"\nconcurrency fuzz case {seed}: KT_FUZZ_ONLY={seed}  shape={kw}  env={env}  variant={variant}  requests={post}")
            raise
    finally:
        stop.set()
        if not hung:  # (a caller still inside the library: leave the engine alone)
            eng.close()


def wide_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930")) + 71
    return [base + 49979687 * i for i in range(int(os.environ.get("KT_FUZZ_WIDE_CASES", "6")))]


@pytest.mark.parametrize("seed", wide_seeds())
def test_random_shapes_with_sums_beyond_int64(seed, oracle_mod, monkeypatch):
    """resource.Quantity never overflows: on a random shape one resource is requested in amounts of 2^59..2^59.6 per pod (a second one
    partly negative), so that `used` of every throttle with a few dozen pods leaves int64 — the two limb scans, the 128-bit join in
    kt_finalize, the wide status the check reads: against the oracle's __int128 arithmetic (test_engine_gpu._wide_parity)."""
    from test_engine_gpu import NOW, _wide_parity, responsible_rows
    kw, env, variant, post = draw_case(seed)
    kw["n_pods"] = min(kw["n_pods"], 4000)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    snap = W.generate(W.small(**kw))
    r = np.random.default_rng(seed ^ 0x1D3)
    first = snap.pod_ctr_off[:snap.n_pods]
    nc = int(snap.pod_ctr_off[snap.n_pods])
    if nc == 0:
        pytest.skip("no containers")
    d0 = int(r.integers(0, snap.D))
    snap.ctr_req[:nc, d0] = 0
    snap.ctr_req[first, d0] = (1 << 59) + r.integers(0, 1 << 58, len(first))  # (the C-ABI takes requests up to 2^60: KT_ERR_OVERFLOW_RISK)
    snap.ctr_present[first] |= 1 << d0
    if snap.D > 1 and r.random() < .5:  # a second resource whose sums are wide AND signed
        d1 = (d0 + 1) % snap.D
        snap.ctr_req[:nc, d1] = 0
        snap.ctr_req[first[::2], d1] = -(1 << 59) - r.integers(0, 1 << 20, len(first[::2]))
        snap.ctr_req[first[1::2], d1] = (1 << 59) + r.integers(0, 1 << 20, len(first[1::2]))
        snap.ctr_present[first] |= 1 << d1
    T = snap.n_thr
    snap.thr_spec.v[:T, d0] = (r.integers(1, 7, T).astype(np.int64) << 60) + 7   # thresholds the wide sums are compared with
    snap.thr_spec.present[:T] |= 1 << d0
    rows = responsible_rows(snap)
    want = oracle_mod.Oracle(snap).reconcile(NOW, rows=rows, wide=True)
    if want.error[:len(rows)].any() or not (want.used_hi[:len(rows)] != (want.used.v[:len(rows)] < 0) * -1).any():
        pytest.skip("no throttle of this shape counts enough pods to leave int64")
    try:
        _wide_parity(snap, oracle_mod, variant & ~E.VARIANT_INCREMENTAL)
    except Exception:
        print(f"\nwide fuzz case {seed}: KT_FUZZ_ONLY={seed}  shape={kw}  env={env}  variant={variant}")
        raise


def shard_seeds():
    only = os.environ.get("KT_FUZZ_ONLY")
    if only:
        return [int(only)]
    base = int(os.environ.get("KT_FUZZ_SEED", "20260930")) + 83
    return [base + 67867967 * i for i in range(int(os.environ.get("KT_FUZZ_SHARD_CASES", "4")))]


@pytest.mark.parametrize("seed", shard_seeds())
def test_random_shapes_in_shards(seed, oracle_mod, monkeypatch):
    """The multi-GPU result on ONE GPU (tests/test_sharded_gpu.py: shards cut as bench.py --gpus N cuts them, partial buffers summed
    as the all-reduce sums them, the sum finalized by every shard's engine, every shard's sweep) on a random shape, 2-8 shards,
    under random switches: every throttle and every pod's summary word against the oracle on the UNSHARDED snapshot."""
    from test_sharded_gpu import _sharded_pipeline
    kw, env, variant, post = draw_case(seed)
    for k in ("n_invalid_pod_sel", "n_invalid_ns_sel", "n_missing_ns"):  # (the pipeline asserts error-free reconciles)
        kw.pop(k, None)
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    world = int(np.random.default_rng(seed ^ 0x5A4D).integers(2, 9))
    try:
        _sharded_pipeline(W.small(**kw), world, oracle_mod, n_thr_sample=kw["n_thr"], n_pod_sample=512, nthreads=4)
    except Exception:
        print(f"\nshard fuzz case {seed}: KT_FUZZ_ONLY={seed}  world={world}  shape={kw}  env={env}")
        raise
