"""The reference's unit-test tables, end to end through the C-ABI: every cluster state that
tests/test_oracle_unit_tables.py builds (resource_amount_test.go, resourcelist_test.go, throttle_selector_test.go,
clusterthrottle_selector_test.go, temporary_threshold_override_test.go, throttle_types_test.go transcriptions, plus
the parity-unpinned selector cases) is replayed on the HIP engine — pod request summation, reconcile, next override,
check with isThrottledOnEqual false and true — and compared bit for bit with the oracle, at every override boundary
of the state -1 s / +0 / +1 s."""
import pytest

from kube_throttler_amd import engine as E
from test_engine_gpu import VARIANTS, VIDS, run_full_parity
from unit_table_states import collect, instants

pytestmark = pytest.mark.gpu


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
def test_override_examples_on_engine(variant):
    """example/throttle-with-temporaryThresholdOverrides.yaml + the ClusterThrottle twin with the example pods, inside
    and outside the override window (expected values hand-traced in tests/examples_overrides.py)."""
    from examples_overrides import run_examples
    from test_engine_gpu import EngineBackend
    run_examples(EngineBackend(variant))
