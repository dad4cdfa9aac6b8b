"""Replays the reference's unit-test tables against the CPU oracle.

Each test cites the reference spec it transcribes.  The reference builds objects with mkPod /
mkNamespace / WithLabels / WithRequests (pkg/apis/schedule/v1alpha1/v1alpha1_suite_test.go:34-75);
the helpers below build the same manifests.
"""
import pytest

from kube_throttler_amd import snapshot as S
from kube_throttler_amd.objects import ClusterState
from kube_throttler_amd.quantity import parse_quantity, parse_rfc3339


RECORDED = None   # tests/test_parity_extended_gpu.py puts a list here to collect every ClusterState the tables build


def _build(cs):
    if RECORDED is not None:
        RECORDED.append(cs)
    return cs.build()


def mk_pod(name, namespace, labels=None, requests=None, init_requests=None, containers=None, overhead=None):
    spec = {"schedulerName": "my-scheduler",
            "containers": containers if containers is not None else [{"name": "ctr", "resources": {"requests": dict(requests or {})}}]}
    if init_requests is not None:
        spec["initContainers"] = [{"name": f"i{k}", "resources": {"requests": r}} for k, r in enumerate(init_requests)]
    if overhead is not None:
        spec["overhead"] = overhead
    return {"kind": "Pod", "metadata": {"name": name, "namespace": namespace, "labels": dict(labels or {})}, "spec": spec}


def amounts(D, dims, counts=None, requests=None):
    a = S.Amounts(1, D)
    a.set_row(0, {dims[k]: int(parse_quantity(v)) for k, v in (requests or {}).items()}, counts)
    return a


# ---------------------------------------------------------------- resource_amount_test.go:27-210
DIMS = {"r1": 0, "r2": 1, "r3": 2}


def is_throttled(oracle_mod, thr, used, on_equal):
    pod, flags = oracle_mod.unit_is_throttled(3, thr, used, on_equal)
    return pod, {k: flags[d] for k, d in DIMS.items() if d in flags}


@pytest.mark.parametrize("b", [False, True])
def test_is_throttled_empty_threshold(oracle_mod, b):
    """resource_amount_test.go:29-59: empty threshold throttles nothing and yields no flags."""
    empty = amounts(3, DIMS)
    assert is_throttled(oracle_mod, empty, amounts(3, DIMS, counts=3), b) == (False, {})
    assert is_throttled(oracle_mod, empty, amounts(3, DIMS, requests={"r1": "1000"}), b) == (False, {})


def test_is_throttled_counts(oracle_mod):
    """resource_amount_test.go:60-113."""
    thr = amounts(3, DIMS, counts=3, requests={"r1": "10", "r2": "20"})
    rr_false = {"r1": False, "r2": False}
    for b in (False, True):
        assert is_throttled(oracle_mod, thr, amounts(3, DIMS, counts=2), b) == (False, rr_false)
    assert is_throttled(oracle_mod, thr, amounts(3, DIMS, counts=3), False) == (False, rr_false)
    assert is_throttled(oracle_mod, thr, amounts(3, DIMS, counts=3), True) == (True, rr_false)
    for b in (False, True):
        assert is_throttled(oracle_mod, thr, amounts(3, DIMS, counts=4), b) == (True, rr_false)


def test_is_throttled_requests(oracle_mod):
    """resource_amount_test.go:114-195."""
    thr = amounts(3, DIMS, counts=3, requests={"r1": "10", "r2": "20"})

    def used(r1, r2):
        return amounts(3, DIMS, requests={"r1": r1, "r2": r2})

    for b in (False, True):
        assert is_throttled(oracle_mod, thr, used("1", "2"), b) == (False, {"r1": False, "r2": False})
    assert is_throttled(oracle_mod, thr, used("10", "20"), False) == (False, {"r1": False, "r2": False})
    assert is_throttled(oracle_mod, thr, used("10", "20"), True) == (False, {"r1": True, "r2": True})
    for b in (False, True):
        assert is_throttled(oracle_mod, thr, used("11", "22"), b) == (False, {"r1": True, "r2": True})
    assert is_throttled(oracle_mod, thr, used("1", "20"), False) == (False, {"r1": False, "r2": False})
    assert is_throttled(oracle_mod, thr, used("1", "20"), True) == (False, {"r1": False, "r2": True})


@pytest.mark.parametrize("b", [False, True])
def test_is_throttled_resource_not_in_threshold(oracle_mod, b):
    """resource_amount_test.go:196-208."""
    thr = amounts(3, DIMS, counts=3, requests={"r1": "10", "r2": "20"})
    assert is_throttled(oracle_mod, thr, amounts(3, DIMS, requests={"r3": "3000"}), b) == \
        (False, {"r1": False, "r2": False})


# ---------------------------------------------------------------- resource_amount_test.go:212-250
def _state_with_pods(pods, extra_dims=("r1", "r2", "r3")):
    cs = ClusterState()
    cs.add_namespace("test")
    # a throttle that names every dimension so that the dimension table is stable
    cs.add({"kind": "Throttle", "metadata": {"name": "dims", "namespace": "test"},
            "spec": {"throttlerName": "other", "threshold": {"resourceRequests": {d: "0" for d in extra_dims}}}})
    for p in pods:
        cs.add(p)
    return _build(cs)


def test_is_throttled_for(oracle_mod):
    b = _state_with_pods([
        mk_pod("p0", "test"),
        mk_pod("p1", "test", requests={"r2": "0"}),
        mk_pod("p2", "test", requests={"r2": "1"}),
        mk_pod("p3", "test", requests={"r1": "1000"}),
        mk_pod("p4", "test", requests={"r3": "1000"}),
    ])
    d = b.dims
    # ResourceCounts throttled => true for any pod (:214-223)
    assert oracle_mod.unit_is_throttled_for(b.snapshot, 0, {}, True)
    flags = {d["r1"]: False, d["r2"]: True}
    # (:224-247)
    assert not oracle_mod.unit_is_throttled_for(b.snapshot, 1, flags, False)  # r2: 0   -> zero request ignored
    assert oracle_mod.unit_is_throttled_for(b.snapshot, 2, flags, False)      # r2: 1
    assert not oracle_mod.unit_is_throttled_for(b.snapshot, 3, flags, False)  # r1 not throttled
    assert not oracle_mod.unit_is_throttled_for(b.snapshot, 4, flags, False)  # r3 has no flag


# ---------------------------------------------------------------- resourcelist_test.go:47-116
def test_pod_request_resource_list_sum(oracle_mod):
    b = _state_with_pods([mk_pod("p", "test", containers=[
        {"name": "a", "resources": {"requests": {"n1": "1"}}}, {"name": "b", "resources": {"requests": {"n1": "1"}}}])],
        extra_dims=("n1", "n2"))
    v, present = oracle_mod.Oracle(b.snapshot).pod_requests()
    assert present[0] == 1 << b.dims["n1"] and v[0, b.dims["n1"]] == 2


def test_pod_request_resource_list_init_containers(oracle_mod):
    """max(max(initContainers), sum(containers)) per resource; n2 only in an init container."""
    b = _state_with_pods([mk_pod("p", "test", containers=[
        {"name": "a", "resources": {"requests": {"n1": "1"}}}, {"name": "b", "resources": {"requests": {"n1": "1"}}}],
        init_requests=[{"n1": "1"}, {"n2": "2"}])], extra_dims=("n1", "n2"))
    v, present = oracle_mod.Oracle(b.snapshot).pod_requests()
    assert present[0] == (1 << b.dims["n1"]) | (1 << b.dims["n2"])
    assert v[0, b.dims["n1"]] == 2 and v[0, b.dims["n2"]] == 2


def test_pod_request_resource_list_overhead_and_zero_keys(oracle_mod):
    """resourcelist.go:27-54,76-84: Add creates keys for zero values; overhead is added on top."""
    b = _state_with_pods([
        mk_pod("p0", "test", requests={"n1": "0"}),
        mk_pod("p1", "test", requests={"n1": "3"}, init_requests=[{"n1": "5", "n2": "0"}], overhead={"n1": "1", "n3": "7"}),
    ], extra_dims=("n1", "n2", "n3"))
    v, present = oracle_mod.Oracle(b.snapshot).pod_requests()
    d = b.dims
    assert present[0] == 1 << d["n1"] and v[0, d["n1"]] == 0
    assert present[1] == (1 << d["n1"]) | (1 << d["n2"]) | (1 << d["n3"])
    assert (v[1, d["n1"]], v[1, d["n2"]], v[1, d["n3"]]) == (6, 0, 7)


# ---------------------------------------------------------------- resourcelist_test.go:119-170 (Add), :320-366 (setMax)
def _used_of(oracle_mod, pods, dims=("n1", "n2", "n3", "n4")):
    """`used` of a throttle selecting every pod of the namespace: the Add fold of reconcile
    (throttle_controller.go:116-119 -> resource_amount.go:91-110 -> resourcelist.go:48-54)."""
    cs = ClusterState()
    cs.add_namespace("test")
    cs.add({"kind": "Throttle", "metadata": {"name": "all", "namespace": "test"},
            "spec": {"throttlerName": "kube-throttler", "selector": {"selectorTerms": [{"podSelector": {}}]},
                     "threshold": {"resourceRequests": {d: "0" for d in dims}}}})
    for p in pods:
        p["spec"]["nodeName"] = "node-1"
        p["status"] = {"phase": "Running"}
        cs.add(p)
    b = _build(cs)
    res = oracle_mod.Oracle(b.snapshot).reconcile(parse_rfc3339("2026-01-01T00:00:00Z"))
    assert not res.error[0]
    return b.amount_to_dict(res.used, 0)


def test_resourcelist_add_merges_key_sets(oracle_mod):
    """resourcelist_test.go:121-143: {n1:0,n2:1,n3:1} + {n2:0,n3:1,n4:2} = {n1:0,n2:1,n3:2,n4:2} — zero-valued keys
    survive on both sides."""
    got = _used_of(oracle_mod, [mk_pod("l", "test", requests={"n1": "0", "n2": "1", "n3": "1"}),
                                mk_pod("r", "test", requests={"n2": "0", "n3": "1", "n4": "2"})])
    assert got == {"resourceCounts": {"pod": 2}, "resourceRequests": {"n1": 0, "n2": 1, "n3": 2, "n4": 2}}


def test_resourcelist_add_equal_key_sets(oracle_mod):
    """resourcelist_test.go:146-170: {0,0,1,1} + {0,1,0,1} = {0,1,1,2}."""
    got = _used_of(oracle_mod, [mk_pod("l", "test", requests={"n1": "0", "n2": "0", "n3": "1", "n4": "1"}),
                                mk_pod("r", "test", requests={"n1": "0", "n2": "1", "n3": "0", "n4": "1"})])
    assert got == {"resourceCounts": {"pod": 2}, "resourceRequests": {"n1": 0, "n2": 1, "n3": 1, "n4": 2}}


def test_resourcelist_set_max_tables(oracle_mod):
    """resourcelist_test.go:321-343 and :345-366 through the one place SetMax is used on the path
    (resourcelist.go:28-38: containers' sum SetMax initContainers' max): lhs = the single container, rhs = the single
    init container.  A key only the rhs has is merged "even if its quantity is zero"."""
    b = _state_with_pods([
        mk_pod("p0", "test", requests={"n1": "1", "n2": "2", "n3": "2"}, init_requests=[{"n2": "1", "n3": "2", "n4": "0"}]),
        mk_pod("p1", "test", requests={"n1": "1", "n2": "1", "n3": "2", "n4": "2"},
               init_requests=[{"n1": "1", "n2": "2", "n3": "1", "n4": "2"}]),
    ], extra_dims=("n1", "n2", "n3", "n4"))
    v, present = oracle_mod.Oracle(b.snapshot).pod_requests()
    d = b.dims
    every = sum(1 << d[k] for k in ("n1", "n2", "n3", "n4"))
    assert present[0] == every and [int(v[0, d[k]]) for k in ("n1", "n2", "n3", "n4")] == [1, 2, 2, 0]
    assert present[1] == every and [int(v[1, d[k]]) for k in ("n1", "n2", "n3", "n4")] == [1, 2, 2, 2]


# ---------------------------------------------------------------- temporary_threshold_override_test.go:27-102
def _override_state(overrides, threshold=None):
    cs = ClusterState()
    cs.add_namespace("default")
    cs.add({"kind": "Throttle", "metadata": {"name": "t", "namespace": "default"},
            "spec": {"throttlerName": "dummy", "threshold": threshold or {},
                     "temporaryThresholdOverrides": overrides}})
    return _build(cs)


BEGIN, END = "2021-08-04T10:00:00Z", "2021-08-05T10:00:00Z"


def _at(text, delta_s=0):
    s, ns = parse_rfc3339(text)
    return (s + delta_s, ns)


def test_override_is_active(oracle_mod):
    zero = (S.ZERO_TIME_S, 0)
    act = oracle_mod.unit_override_is_active
    b = _override_state([{}, {"begin": BEGIN}, {"end": END}, {"begin": BEGIN, "end": END},
                         {"begin": "not-time"}, {"end": "not-time"}])
    s = b.snapshot
    # Empty Begin/End: always active (:40-47)
    assert act(s, 0, zero) == 1 and act(s, 0, _at(BEGIN)) == 1 and act(s, 0, _at(END)) == 1
    # Begin only (:48-58)
    assert act(s, 1, _at(BEGIN, -1)) == 0 and act(s, 1, _at(BEGIN)) == 1 and act(s, 1, _at(BEGIN, 1)) == 1
    assert act(s, 1, _at(BEGIN, 65535 * 3600)) == 1
    # End only (:59-69)
    assert act(s, 2, _at(END, -65535 * 3600)) == 1 and act(s, 2, _at(END, -1)) == 1
    assert act(s, 2, _at(END)) == 1 and act(s, 2, _at(END, 1)) == 0
    # Begin and End (:70-83)
    assert [act(s, 3, _at(BEGIN, -1)), act(s, 3, _at(BEGIN)), act(s, 3, _at(BEGIN, 1)), act(s, 3, _at(END, -1)),
            act(s, 3, _at(END)), act(s, 3, _at(END, 1))] == [0, 1, 1, 1, 1, 0]
    # parse failures raise (:84-99)
    assert act(s, 4, zero) == -1 and act(s, 5, zero) == -1


# ---------------------------------------------------------------- throttle_types_test.go:31-152
NOW = "2006-01-02T15:04:05Z"


def _fmt(delta):
    import datetime as dt
    t = dt.datetime(2006, 1, 2, 15, 4, 5, tzinfo=dt.timezone.utc) + dt.timedelta(seconds=delta)
    return t.strftime("%Y-%m-%dT%H:%M:%SZ")


THRESHOLD = {"resourceCounts": {"pod": 0}, "resourceRequests": {"cpu": "1"}}
OVERRIDE1 = {"begin": _fmt(-60), "end": _fmt(60),
             "threshold": {"resourceCounts": {"pod": 2}, "resourceRequests": {"cpu": "2"}}}
OVERRIDE2 = {"begin": _fmt(-60), "end": _fmt(60),
             "threshold": {"resourceCounts": {"pod": 3}, "resourceRequests": {"cpu": "3", "memory": "3"}}}
ERRORED = {"begin": "error", "end": "error"}


def _calc(oracle_mod, overrides):
    b = _override_state(overrides, THRESHOLD)
    # make sure both cpu and memory are dimensions even when no override names memory
    out, any_err = oracle_mod.unit_calculate_threshold(b.snapshot, 0, parse_rfc3339(NOW))
    return b.amount_to_dict(out, 0), any_err, b


def test_calculate_threshold_none(oracle_mod):
    got, err, _ = _calc(oracle_mod, [])
    assert got == {"resourceCounts": {"pod": 0}, "resourceRequests": {"cpu": 1}} and not err


def test_calculate_threshold_single_active(oracle_mod):
    got, err, _ = _calc(oracle_mod, [OVERRIDE1])
    assert got == {"resourceCounts": {"pod": 2}, "resourceRequests": {"cpu": 2}} and not err


def test_calculate_threshold_merged_first_wins(oracle_mod):
    got, err, _ = _calc(oracle_mod, [OVERRIDE1, OVERRIDE2])
    assert got == {"resourceCounts": {"pod": 2}, "resourceRequests": {"cpu": 2, "memory": 3}} and not err


def test_calculate_threshold_errored_override_skipped(oracle_mod):
    got, err, b = _calc(oracle_mod, [OVERRIDE1, ERRORED])
    assert got == {"resourceCounts": {"pod": 2}, "resourceRequests": {"cpu": 2}} and err
    # throttle_types_test.go:146-148 pins the message prefix "index 1: Failed to parse Begin: "
    assert len(b.thr_messages[0]) == 1 and b.thr_messages[0][0].startswith("index 1: Failed to parse Begin: parsing time \"error\"")


def test_calculate_threshold_override_replaces_not_overlays(oracle_mod):
    """throttle_types.go:96-98: an active override REPLACES the threshold; resources it does not name
    become un-thresholded (no resourceCounts here)."""
    got, err, _ = _calc(oracle_mod, [{"begin": _fmt(-60), "threshold": {"resourceRequests": {"memory": "5"}}}])
    assert got == {"resourceRequests": {"memory": 5}} and not err


def test_readme_override_example_follows_the_code(oracle_mod):
    """README.md:157-198 (the documentation example of temporaryThresholdOverrides).  At 2019-02-16T00:00:00+09:00 both
    overrides are active and merge first-wins: cpu "5" from [0], memory "8Gi" from [1] — as the README says.  The README
    also keeps `resourceCounts: pod: 3` ("this is not overridden") and is silent about nvidia.com/gpu; the CODE replaces
    the whole threshold by the merged overrides (throttle_types.go:96-98, pinned by throttle_types_test.go:101-127), so
    neither the pod count nor the gpu limit survives while an override is active.  The engine follows the code."""
    threshold = {"resourceCounts": {"pod": 3}, "resourceRequests": {"cpu": "200m", "memory": "1Gi", "nvidia.com/gpu": "2"}}
    overrides = [
        {"begin": "2019-02-01T00:00:00+09:00", "end": "2019-03-01T00:00:00+09:00", "threshold": {"resourceRequests": {"cpu": "5"}}},
        {"begin": "2019-02-15T00:00:00+09:00", "end": "2019-03-01T00:00:00+09:00",
         "threshold": {"resourceRequests": {"cpu": "1", "memory": "8Gi"}}}]
    b = _override_state(overrides, threshold)

    def at(text):
        out, err = oracle_mod.unit_calculate_threshold(b.snapshot, 0, parse_rfc3339(text))
        assert not err
        return b.amount_to_dict(out, 0)

    gi = 1 << 30
    assert at("2019-02-16T00:00:00+09:00") == {"resourceRequests": {"cpu": 5, "memory": 8 * gi}}
    assert at("2019-02-10T00:00:00+09:00") == {"resourceRequests": {"cpu": 5}}                 # only [0] active
    assert at("2019-03-01T00:00:00+09:00") == {"resourceRequests": {"cpu": 5, "memory": 8 * gi}}   # end is inclusive
    base = {"resourceCounts": {"pod": 3}, "resourceRequests": {"cpu": parse_quantity("200m"), "memory": gi, "nvidia.com/gpu": 2}}
    assert at("2019-03-01T00:00:01+09:00") == base and at("2019-01-31T23:59:59+09:00") == base


# ---------------------------------------------------------------- throttle_selector_test.go / clusterthrottle_selector_test.go
def _selector_state(kind, terms, pods, namespaces):
    cs = ClusterState()
    for n, l in namespaces:
        cs.add_namespace(n, l)
    md = {"name": "t"}
    if kind == "Throttle":
        md["namespace"] = pods[0]["metadata"]["namespace"]
    cs.add({"kind": kind, "metadata": md, "spec": {"throttlerName": "kube-throttler", "selector": {"selectorTerms": terms}}})
    for p in pods:
        cs.add(p)
    return _build(cs)


def test_throttle_selector_empty_matches_nothing(oracle_mod):
    """throttle_selector_test.go:29-42."""
    b = _selector_state("Throttle", [], [mk_pod("test", "test", {"test": "test"})], [("test", {})])
    assert oracle_mod.unit_selector_matches(b.snapshot, 0, 0) == 0


def test_throttle_selector_terms_are_ored(oracle_mod):
    """throttle_selector_test.go:43-81."""
    terms = [{"podSelector": {"matchLabels": {"test1": "test1"}}}, {"podSelector": {"matchLabels": {"test2": "test2"}}}]
    pods = [mk_pod("test1", "ns", {"test1": "test1"}), mk_pod("test2", "ns", {"test2": "test2"}),
            mk_pod("test3", "ns", {"test1": "test2"})]
    b = _selector_state("Throttle", terms, pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(3)] == [1, 1, 0]


def test_throttle_selector_empty_term_matches_everything(oracle_mod):
    """throttle_selector_test.go:84-102."""
    pods = [mk_pod("test1", "ns", {"test": "test"}), mk_pod("test2", "ns")]
    b = _selector_state("Throttle", [{}], pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [1, 1]


def test_clusterthrottle_selector_empty_matches_nothing(oracle_mod):
    """clusterthrottle_selector_test.go:29-42."""
    b = _selector_state("ClusterThrottle", [], [mk_pod("test", "test", {"test": "test"})], [("test", {"test": "test"})])
    assert oracle_mod.unit_selector_matches(b.snapshot, 0, 0) == 0


def test_clusterthrottle_selector_terms_are_ored(oracle_mod):
    """clusterthrottle_selector_test.go:43-89: namespaceSelector AND podSelector per term, OR over terms."""
    t1, t2 = {"test1": "test1"}, {"test2": "test2"}
    terms = [{"namespaceSelector": {"matchLabels": t1}, "podSelector": {"matchLabels": t1}},
             {"namespaceSelector": {"matchLabels": t2}, "podSelector": {"matchLabels": t2}}]
    pods = [mk_pod("test1", "test1", t1), mk_pod("test2", "test2", t2), mk_pod("test3", "test1", t2)]
    b = _selector_state("ClusterThrottle", terms, pods, [("test1", t1), ("test2", t2)])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(3)] == [1, 1, 0]


def test_clusterthrottle_selector_empty_term_matches_everything(oracle_mod):
    """clusterthrottle_selector_test.go:92-110."""
    pods = [mk_pod("test1", "test1", {"test": "test"}), mk_pod("test2", "test2")]
    b = _selector_state("ClusterThrottle", [{}], pods, [("test1", {"test": "test"}), ("test2", None)])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [1, 1]


# ---------------------------------------------------------------- parity-unpinned L0 behaviour (restated)
def test_match_expressions_operators(oracle_mod):
    terms = [{"podSelector": {"matchExpressions": [
        {"key": "tier", "operator": "In", "values": ["a", "b"]},
        {"key": "env", "operator": "NotIn", "values": ["dev"]},
        {"key": "owner", "operator": "Exists"},
        {"key": "legacy", "operator": "DoesNotExist"}]}}]
    pods = [
        mk_pod("ok", "ns", {"tier": "a", "owner": "x"}),                     # env absent => NotIn true
        mk_pod("ok2", "ns", {"tier": "b", "env": "prod", "owner": ""}),
        mk_pod("bad-tier", "ns", {"tier": "c", "owner": "x"}),
        mk_pod("no-tier", "ns", {"owner": "x"}),
        mk_pod("env-dev", "ns", {"tier": "a", "env": "dev", "owner": "x"}),
        mk_pod("no-owner", "ns", {"tier": "a"}),
        mk_pod("legacy", "ns", {"tier": "a", "owner": "x", "legacy": "1"}),
    ]
    b = _selector_state("Throttle", terms, pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(7)] == [1, 1, 0, 0, 0, 0, 0]


def test_invalid_selectors(oracle_mod):
    """podSelector conversion error propagates (throttle_selector.go:48-52); an earlier matching term wins
    first (:32-40); namespaceSelector errors are swallowed to "no match" (clusterthrottle_selector.go:63-69)."""
    bad = {"podSelector": {"matchExpressions": [{"key": "a", "operator": "In", "values": []}]}}
    good = {"podSelector": {"matchLabels": {"app": "x"}}}
    pods = [mk_pod("p0", "ns", {"app": "x"}), mk_pod("p1", "ns", {"app": "y"})]
    b = _selector_state("Throttle", [good, bad], pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [1, -1]
    b = _selector_state("Throttle", [bad, good], pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [-1, -1]
    bad_ns = {"namespaceSelector": {"matchExpressions": [{"key": "a", "operator": "Bogus"}]},
              "podSelector": {"matchLabels": {"app": "x"}}}
    b = _selector_state("ClusterThrottle", [bad_ns], pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [0, 0]
    # bad podSelector behind a non-matching namespaceSelector is never converted (clusterthrottle_selector.go:72-79)
    hidden = {"namespaceSelector": {"matchLabels": {"zone": "z"}}, "podSelector": bad["podSelector"]}
    b = _selector_state("ClusterThrottle", [hidden], pods, [("ns", {})])
    assert [oracle_mod.unit_selector_matches(b.snapshot, 0, i) for i in range(2)] == [0, 0]
