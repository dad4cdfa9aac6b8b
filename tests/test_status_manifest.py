"""Status write-back: the `status` document UpdateStatus would persist, built from a reconcile
result.  The reconcile here comes from the CPU oracle (checker only); the GPU suite runs the same scenarios through the
engine (tests/test_engine_gpu.py scenario tests use the same write_back_status)."""
from scenario_runner import OracleBackend, build_cluster_state, load_scenarios

from kube_throttler_amd.quantity import parse_rfc3339


def _scenario(name):
    return next(s for s in load_scenarios() if s["name"] == name)


def test_status_document_many_pods(oracle_mod):
    """test/integration/throttle_test.go:167-197: 20 x 50m against threshold cpu=1."""
    sc = _scenario("G1-vi-many-pods")
    cs = build_cluster_state(sc)
    for p in cs.pods:
        if p["metadata"]["name"] != "pod-20":
            assert p["spec"].get("nodeName")
    built = cs.build()
    res = OracleBackend(oracle_mod).reconcile(built, parse_rfc3339(sc["now"]))
    st = built.status_manifest(res, 0, sc["now"])
    assert st == {
        "used": {"resourceCounts": {"pod": 20}, "resourceRequests": {"cpu": "1"}},
        "calculatedThreshold": {"threshold": {"resourceRequests": {"cpu": "1"}}, "calculatedAt": sc["now"], "messages": []},
        "throttled": {"resourceCounts": {"pod": False}, "resourceRequests": {"cpu": True}},
    }
    # written back and reconciled again: nothing is replaced, calculatedAt keeps its first value
    cs.throttles[0]["status"] = st
    built = cs.build()
    res = OracleBackend(oracle_mod).reconcile(built, parse_rfc3339("2030-01-01T00:00:00Z"))
    assert not res.calc_updated[0]
    again = built.status_manifest(res, 0, "2030-01-01T00:00:00Z", previous=st)
    assert again == st


def test_status_document_binary_si(oracle_mod):
    """Memory written in binary SI stays in binary SI; an empty throttle writes `used: {}`."""
    sc = dict(_scenario("G1-i-within-threshold"))
    sc["throttles"] = [{
        "kind": "Throttle", "metadata": {"name": "mem", "namespace": "default"},
        "spec": {"throttlerName": "kube-throttler",
                 "selector": {"selectorTerms": [{"podSelector": {"matchLabels": {"throttle": "mem"}}}]},
                 "threshold": {"resourceCounts": {"pod": 3}, "resourceRequests": {"memory": "2Gi", "cpu": "1500m"}}}},
        {"kind": "Throttle", "metadata": {"name": "idle", "namespace": "default"},
         "spec": {"throttlerName": "kube-throttler",
                  "selector": {"selectorTerms": [{"podSelector": {"matchLabels": {"throttle": "nobody"}}}]},
                  "threshold": {"resourceRequests": {"memory": "1Gi"}}}}]
    sc["pods"] = [{"name": f"p{i}", "namespace": "default", "labels": {"throttle": "mem"}, "scheduled": True,
                   "requests": {"memory": "768Mi", "cpu": "250m"}} for i in range(2)]
    sc.pop("pods_repeat", None)
    cs = build_cluster_state(sc)
    built = cs.build()
    res = OracleBackend(oracle_mod).reconcile(built, parse_rfc3339(sc["now"]))
    by_name = {n: built.status_manifest(res, i, sc["now"]) for i, n in enumerate(built.thr_names)}
    mem = by_name["default/mem"]
    assert mem["used"] == {"resourceCounts": {"pod": 2}, "resourceRequests": {"memory": "1536Mi", "cpu": "500m"}}
    assert mem["calculatedThreshold"]["threshold"] == {"resourceCounts": {"pod": 3},
                                                       "resourceRequests": {"memory": "2Gi", "cpu": "1500m"}}
    assert mem["throttled"] == {"resourceCounts": {"pod": False}, "resourceRequests": {"memory": False, "cpu": False}}
    idle = by_name["default/idle"]
    assert idle["used"] == {}
    assert idle["throttled"] == {"resourceCounts": {"pod": False}, "resourceRequests": {"memory": False}}
