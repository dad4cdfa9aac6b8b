"""Prometheus gauges fed from the ENGINE's reconcile result: the exposition text must be identical to
the one produced from the CPU oracle's result on the same cluster — including the corner rows of the reference's
recorders (throttle_metrics.go:112-130, metrics_recorder.go:37-46): nil resourceCounts export 0, a throttle no pod
matches exports no `used` requests, cpu in milli-units."""
import numpy as np
import pytest

from kube_throttler_amd import engine as E
from kube_throttler_amd.metrics import MetricsRecorder
from kube_throttler_amd.objects import ClusterState

pytestmark = pytest.mark.gpu
NOW = (1767225600, 0)


def _cluster():
    cs = ClusterState()
    for ns, team in (("default", "a"), ("batch", "b"), ("empty", "a")):
        cs.add_namespace(ns, {"kubernetes.io/metadata.name": ns, "team": team})
    both = {"throttlerName": "kube-throttler", "threshold": {"resourceCounts": {"pod": 3},
                                                              "resourceRequests": {"cpu": "1", "memory": "1Gi"}}}
    only_requests = {"throttlerName": "kube-throttler", "threshold": {"resourceRequests": {"cpu": "250m", "nvidia.com/gpu": "2"}}}
    only_counts = {"throttlerName": "kube-throttler", "threshold": {"resourceCounts": {"pod": 1}}}
    sel = lambda labels: {"selectorTerms": [{"podSelector": {"matchLabels": labels}}]}
    cs.add({"kind": "Throttle", "metadata": {"namespace": "default", "name": "t-both", "uid": "u-1"}, "spec": dict(both, selector=sel({"app": "web"}))})
    cs.add({"kind": "Throttle", "metadata": {"namespace": "batch", "name": "t-req", "uid": "u-2"}, "spec": dict(only_requests, selector=sel({"app": "job"}))})
    cs.add({"kind": "Throttle", "metadata": {"namespace": "empty", "name": "t-nomatch", "uid": "u-3"}, "spec": dict(both, selector=sel({"app": "none"}))})
    cs.add({"kind": "Throttle", "metadata": {"namespace": "default", "name": "t-other", "uid": "u-4"},
            "spec": dict(only_counts, throttlerName="someone-else", selector=sel({"app": "web"}))})
    cs.add({"kind": "ClusterThrottle", "metadata": {"name": "c-team-a", "uid": "u-5"},
            "spec": dict(only_counts, selector={"selectorTerms": [{"podSelector": {"matchExpressions": [{"key": "app", "operator": "Exists"}]},
                                                                    "namespaceSelector": {"matchLabels": {"team": "a"}}}]})})
    cs.add({"kind": "ClusterThrottle", "metadata": {"name": "c-all", "uid": "u-6"},
            "spec": dict(both, selector={"selectorTerms": [{"podSelector": {}, "namespaceSelector": {}}]})})
    pods = (("default", "w0", "web", "500m", "256Mi", None), ("default", "w1", "web", "700m", "512Mi", None),
            ("batch", "j0", "job", "300m", None, "1"), ("batch", "j1", "job", "0", "64Mi", "2"), ("default", "x0", None, "100m", "1Mi", None))
    for ns, name, app, cpu, mem, gpu in pods:
        req = {"cpu": cpu}
        if mem:
            req["memory"] = mem
        if gpu:
            req["nvidia.com/gpu"] = gpu
        cs.add({"kind": "Pod", "metadata": {"namespace": ns, "name": name, "labels": {"app": app} if app else {}},
                "spec": {"schedulerName": "my-scheduler", "nodeName": "node-1", "containers": [{"resources": {"requests": req}}]},
                "status": {"phase": "Running"}})
    return cs.build()


@pytest.mark.parametrize("variant", [E.VARIANT_INDEXED, E.VARIANT_DENSE], ids=["indexed", "dense"])
def test_exposition_from_the_engine_equals_the_oracle_fed_one(variant, oracle_mod):
    built = _cluster()
    want_rec = oracle_mod.Oracle(built.snapshot).reconcile(NOW)
    eng = E.Engine.for_snapshot(built.snapshot, variant)
    try:
        got_rec = eng.reconcile(NOW, apply=False)
    finally:
        eng.close()
    got = MetricsRecorder().record(built, got_rec).exposition()
    want = MetricsRecorder().record(built, want_rec).exposition()
    assert got == want
    samples = {line.split(" ")[0]: float(line.split(" ")[1]) for line in got.splitlines() if line and line[0] != "#"}

    def s(metric, **lab):
        return samples[metric + "{" + ",".join(f'{k}="{v}"' for k, v in sorted(lab.items())) + "}"]

    # spot values computed by hand: used of t-both = w0 + w1; nil resourceCounts export 0; cpu in milli-units
    t = dict(namespace="default", name="t-both", uid="u-1")
    assert s("throttle_status_used_resourceRequests", resource="cpu", **t) == 1200
    assert s("throttle_status_used_resourceCounts", resource="pod", **t) == 2
    assert s("throttle_status_throttled_resourceRequests", resource="cpu", **t) == 1
    r = dict(namespace="batch", name="t-req", uid="u-2")
    assert s("throttle_spec_threshold_resourceCounts", resource="pod", **r) == 0        # nil counts export 0
    assert s("throttle_status_used_resourceRequests", resource="nvidia.com/gpu", **r) == 3
    assert s("throttle_status_throttled_resourceRequests", resource="nvidia.com/gpu", **r) == 1
    n = dict(namespace="empty", name="t-nomatch", uid="u-3")
    assert s("throttle_status_used_resourceCounts", resource="pod", **n) == 0
    assert not any(k.startswith("throttle_status_used_resourceRequests{") and 'name="t-nomatch"' in k for k in samples)
    assert not any('name="t-other"' in k for k in samples)   # another throttler's throttle is never reconciled, never recorded
    assert s("clusterthrottle_status_used_resourceCounts", resource="pod", name="c-team-a", uid="u-5") == 2
    assert s("clusterthrottle_status_used_resourceCounts", resource="pod", name="c-all", uid="u-6") == 5
