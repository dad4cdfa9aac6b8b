"""GPU parity beyond tests/test_engine_gpu.py (this file sorts last on purpose: the long-standing suite runs first).

The reference's unit-test tables, end to end through the C-ABI: every cluster state that
tests/test_oracle_unit_tables.py builds (resource_amount_test.go, resourcelist_test.go, throttle_selector_test.go,
clusterthrottle_selector_test.go, temporary_threshold_override_test.go, throttle_types_test.go transcriptions, plus
the parity-unpinned selector cases) is replayed on the HIP engine — pod request summation, reconcile, next override,
check with isThrottledOnEqual false and true — and compared bit for bit with the oracle, at every override boundary
of the state -1 s / +0 / +1 s.  Also here: the random manifest-level clusters, a cluster without pods, the override
example manifests, concurrent callers of one engine, and the newer scenarios of the C++ plugin mirror."""
import os
import subprocess

import numpy as np
import pytest

from kube_throttler_amd import engine as E
from kube_throttler_amd import snapshot as S
from kube_throttler_amd import workload as W
from test_engine_gpu import NOW, VARIANTS, VIDS, _stored_status, responsible_rows, run_full_parity
from unit_table_states import collect, instants

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_override_examples_on_engine(variant):
    """example/throttle-with-temporaryThresholdOverrides.yaml + the ClusterThrottle twin with the example pods, inside
    and outside the override window (expected values hand-traced in tests/examples_overrides.py)."""
    from examples_overrides import run_examples
    from test_engine_gpu import EngineBackend
    run_examples(EngineBackend(variant))


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_random_manifest_clusters_on_engine(variant, oracle_mod):
    """The random manifest-level clusters of tests/test_manifest_model.py (where the oracle is pinned against an
    independent manifest-level model) through the engine: snapshots built by objects.py rather than by the workload
    generator — other label / selector / override / missing-namespace mixes, resourceCounts-only thresholds, zero
    thresholds."""
    from test_manifest_model import random_cluster
    from kube_throttler_amd.quantity import parse_rfc3339
    for seed in range(32):
        for now_text in ("2026-01-01T00:00:00Z", "2026-01-20T00:00:00Z"):
            snap = random_cluster(seed).build().snapshot
            try:
                run_full_parity(snap, oracle_mod, variant, now=parse_rfc3339(now_text), nthreads=1)
            except AssertionError as ex:
                raise AssertionError(f"random cluster seed {seed} at {now_text}: {ex}") from ex


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_unit_table_states_on_engine(variant, oracle_mod):
    states = collect(oracle_mod)
    assert len(states) >= 30
    runs = 0
    for label, cs in states:
        for now in instants(cs):
            snap = cs.build().snapshot   # fresh: run_full_parity stores the reconciled status into it
            try:
                run_full_parity(snap, oracle_mod, variant, now=now, nthreads=1)
            except AssertionError as ex:
                raise AssertionError(f"state {label} at {now}: {ex}") from ex
            runs += 1
    assert runs >= len(states)


def test_host_plugin_extended_scenarios():
    """The C++ plugin mirror's newer scenarios (tests/cpp/host_plugin_test.cpp `extended`): status write-back with
    UpdateStatus change detection and canonical quantities, the pod Update / Delete handlers' reservation moves."""
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kube_throttler_amd", "host")
    subprocess.check_call(["make", "-C", host, "host_plugin_test"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(host, "host_plugin_test"), "extended"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all expectations held" in r.stdout


def test_state_without_pods(oracle_mod):
    """A throttle with overrides and not a single pod in the cluster: every launch of the path has zero pod rows."""
    from test_oracle_unit_tables import OVERRIDE1, THRESHOLD
    from kube_throttler_amd.objects import ClusterState
    from kube_throttler_amd.quantity import parse_rfc3339
    cs = ClusterState()
    cs.add_namespace("default")
    cs.add({"kind": "Throttle", "metadata": {"name": "t", "namespace": "default"},
            "spec": {"throttlerName": cs.throttler_name, "threshold": THRESHOLD, "temporaryThresholdOverrides": [OVERRIDE1],
                     "selector": {"selectorTerms": [{"podSelector": {}}]}}})
    snap = cs.build().snapshot
    assert snap.n_pods == 0
    _, _, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED, now=parse_rfc3339("2006-01-02T15:04:05Z"), nthreads=1)
    assert rec.calc_updated[0] and not rec.used.has_count[0]


def test_concurrent_callers(oracle_mod):
    """Thread safety at the C-ABI (SURVEY.md 8b "Threading"; the reference hammers its reserved cache from 2000
    goroutines, pkg/controllers/reserved_resource_amounts_test.go:44-108): writer threads replace reservations on their
    own throttle rows while other threads check (kt_check = launch + fetch as one critical section), read the
    reservations back and run dry reconciles.  No call fails, every atomic check is self-consistent (summary counters
    = its own status matrix), and the end state is the serial one: parity with the oracle."""
    import threading
    snap = W.generate(W.small(seed=33, n_pods=800, n_thr=40, n_cluster=20))
    _stored_status(snap, oracle_mod)
    eng = E.Engine.for_snapshot(snap)
    rows = responsible_rows(snap)[:12]
    n_writers, rounds = 4, 20
    errors = []

    def amount(j, it):
        a = S.Amounts(1, snap.D)
        last = it == rounds - 1
        a.set_row(0, {0: 1000 * (j + 1) + (0 if last else it + 1), 1: (1 << 30) if last else (it + 1) << 20},
                  count=(j % 3 + 1) if last else it + 1)
        return a

    def guarded(fn):
        def run(*args):
            try:
                fn(*args)
            except Exception as ex:  # noqa: BLE001 - collected and re-raised on the main thread
                errors.append(ex)
        return run

    @guarded
    def writer(k):
        for it in range(rounds):
            for j, t in enumerate(rows):
                if j % n_writers == k:
                    eng.set_reserved(np.array([t], np.int32), amount(j, it))

    @guarded
    def checker(step):
        sample = np.arange(step, snap.n_pods, 7, dtype=np.int64)
        for _ in range(rounds):
            st, sm = eng.check_atomic(rows=sample, want_status=True)
            verdict, n_exc, n_act, n_ins = S.summary_fields(sm)
            ok = verdict != S.VERDICT_ERROR
            np.testing.assert_array_equal(n_exc[ok], (st == S.EXCEEDS).sum(axis=1)[ok])
            np.testing.assert_array_equal(n_act[ok], (st == S.ACTIVE).sum(axis=1)[ok])
            np.testing.assert_array_equal(n_ins[ok], (st == S.INSUFFICIENT).sum(axis=1)[ok])
            blocked = (n_exc + n_act + n_ins) > 0
            np.testing.assert_array_equal(verdict[ok] == S.VERDICT_BLOCK, blocked[ok])

    @guarded
    def reader():
        for _ in range(rounds):
            eng.fetch_reserved()
            eng.reconcile(NOW, apply=False)

    try:
        threads = [threading.Thread(target=writer, args=(k,), daemon=True) for k in range(n_writers)]
        threads += [threading.Thread(target=checker, args=(k,), daemon=True) for k in range(2)]
        threads.append(threading.Thread(target=reader, daemon=True))
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=240)     # a deadlock fails the test instead of hanging the box
        assert not any(th.is_alive() for th in threads), "engine calls did not return: deadlock?"
        if errors:
            raise errors[0]
        for j, t in enumerate(rows):
            a = amount(j, rounds - 1)
            for f in ("v", "present", "count", "has_count"):
                getattr(snap.thr_reserved, f)[t] = getattr(a, f)[0]
        res = eng.fetch_reserved()
        T = snap.n_thr
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(res, f)[:T], getattr(snap.thr_reserved, f)[:T], err_msg=f)
        o = oracle_mod.Oracle(snap)
        st_w, sm_w = o.check()
        st_g, sm_g = eng.check_atomic(n=snap.n_pods, want_status=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()


def test_reconcile_of_single_keys(oracle_mod):
    """The reference reconciles ONE throttle per workqueue key (throttle_controller.go:84-133): kt_reconcile_rows_launch
    renews only the listed rows — the others keep their stored status — and the PreFilter that follows sees exactly the
    mix of renewed and stored statuses the oracle sees after the same keys were written back."""
    from test_engine_gpu import NOW, assert_reconcile_equal, responsible_rows
    snap = W.generate(W.small(seed=61, n_pods=2500, n_thr=80, n_cluster=40))
    T = snap.n_thr
    o = oracle_mod.Oracle(snap)
    eng = E.Engine.for_snapshot(snap)
    try:
        resp = responsible_rows(snap)
        keys = resp[::3]                      # every third responsible throttle is reconciled
        want = o.reconcile(NOW, rows=keys)
        got_all = eng.reconcile_rows(NOW, keys, apply=True)
        got = E.ReconcileResult(len(keys), snap.D)
        for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
            getattr(got, name)[:len(keys)] = getattr(got_all, name)[keys]
        for tab in ("used", "calc"):
            for f in ("v", "present", "count", "has_count"):
                getattr(getattr(got, tab), f)[:len(keys)] = getattr(getattr(got_all, tab), f)[keys]
        assert_reconcile_equal(got, want, len(keys))
        # the rows that were not keys report the status they had
        others = np.setdiff1d(np.arange(T), keys)
        np.testing.assert_array_equal(got_all.used.v[others], snap.thr_used.v[others])
        np.testing.assert_array_equal(got_all.used.present[others], snap.thr_used.present[others])
        np.testing.assert_array_equal(got_all.calc_updated[others], 0)
        # UpdateStatus of the keys on the oracle side, then PreFilter against the mixed statuses
        snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=keys)
        for on_equal in (False, True):
            st_w, sm_w = o.check(on_equal=on_equal, nthreads=8)
            st_g, sm_g = eng.check(n=snap.n_pods, on_equal=on_equal, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()
