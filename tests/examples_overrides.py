"""The reference's remaining example manifests — example/throttle-with-temporaryThresholdOverrides.yaml and
example/clthrottle-with-temporaryThresholdOverrides.yaml (fixture: tests/golden/example_config0.json) — with the example
pods, reconciled and checked inside and outside the override window.  Expected values are hand-traced from
throttle_types.go:65-106,128-153 / clusterthrottle_types.go:30-55 (see the comments); `backend` is the oracle on the CPU
suite and the HIP engine on the GPU suite (tests/scenario_runner.py backends)."""
import copy
import json
import os

from kube_throttler_amd import snapshot as S
from kube_throttler_amd.objects import ClusterState
from kube_throttler_amd.quantity import parse_quantity, parse_rfc3339
from scenario_runner import write_back_status

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {S.NOT_THROTTLED: "not-throttled", S.ACTIVE: "active", S.INSUFFICIENT: "insufficient",
         S.EXCEEDS: "pod-requests-exceeds-threshold"}


def build_state() -> ClusterState:
    with open(os.path.join(GOLDEN, "example_config0.json")) as fh:
        ex = json.load(fh)["manifests"]
    cs = ClusterState()
    cs.add_namespace("default", {"throttle": "true"})
    cs.add_namespace("other", {})
    t = copy.deepcopy(ex["throttle-with-temporaryThresholdOverrides"])
    t["metadata"]["namespace"] = "default"
    cs.add(t)
    cs.add(copy.deepcopy(ex["clthrottle-with-temporaryThresholdOverrides"]))
    for name, ns, bound in (("pod1", "default", True), ("pod2", "default", True), ("pod1m", "default", True),
                            ("pod3", "default", False), ("pod3", "other", False), ("pod1m", "default", False)):
        p = copy.deepcopy(ex[name])
        p["metadata"]["namespace"] = ns
        p["metadata"]["name"] = f"{name}-{ns}-{'bound' if bound else 'pending'}"
        p["status"] = {"phase": "Running" if bound else "Pending"}
        if bound:
            p["spec"]["nodeName"] = "node-1"
        cs.add(p)
    return cs


def run_examples(backend):
    gi = 1 << 30
    used = {"resourceCounts": {"pod": 3}, "resourceRequests": {"cpu": parse_quantity("500m"), "memory": gi // 2}}
    cases = [
        # inside the window [2019-02-01, 2019-03-01] (+09:00): the override REPLACES the threshold -> cpu 5 only
        ("2019-02-16T00:00:00+09:00", {"resourceRequests": {"cpu": 5}}, (False, {"cpu": False}),
         {"pod3-default-pending": ({"default/t1": "not-throttled", "/ct1": "not-throttled"},) * 2,
          "pod3-other-pending": ({}, {}),
          "pod1m-default-pending": ({"default/t1": "not-throttled", "/ct1": "not-throttled"},) * 2}),
        # outside: pod 5 / cpu 200m / memory 1Gi.  used cpu 500m >= 200m -> throttled.cpu; a 300m pod alone exceeds
        # 200m; the 512Mi pod fits exactly: 512Mi + 512Mi = 1Gi is "insufficient" only when isThrottledOnEqual
        ("2026-01-01T00:00:00Z", {"resourceCounts": {"pod": 5}, "resourceRequests": {"cpu": parse_quantity("200m"), "memory": gi}},
         (False, {"cpu": True, "memory": False}),
         {"pod3-default-pending": ({"default/t1": "pod-requests-exceeds-threshold", "/ct1": "pod-requests-exceeds-threshold"},) * 2,
          "pod3-other-pending": ({}, {}),
          "pod1m-default-pending": ({"default/t1": "not-throttled", "/ct1": "not-throttled"},
                                    {"default/t1": "insufficient", "/ct1": "insufficient"})}),
    ]
    for now_text, want_calc, (want_pod, want_flags), want_checks in cases:
        cs = build_state()
        now = parse_rfc3339(now_text)
        built = cs.build()
        res = backend.reconcile(built, now)
        for i, nn in enumerate(built.thr_names):
            where = f"{nn} at {now_text}"
            assert not res.error[i], where
            assert built.amount_to_dict(res.used, i) == used, f"{where}: used"
            assert built.amount_to_dict(res.calc, i) == want_calc, f"{where}: calculated threshold"
            assert bool(res.thrl_pod[i]) == want_pod, f"{where}: throttled.pod"
            got = {name: bool(int(res.thrl_flag[i]) >> d & 1) for name, d in built.dims.items() if int(res.thrl_has[i]) >> d & 1}
            assert got == want_flags, f"{where}: throttled.resourceRequests {got}"
        write_back_status(cs, built, res, now_text)
        assert cs.throttles[0]["status"]["used"]["resourceRequests"] == {"cpu": "500m", "memory": "512Mi"}
        built = cs.build()
        names = [p["metadata"]["name"] for p in cs.pods]
        for k, on_equal in enumerate((False, True)):
            rows = [names.index(n) for n in want_checks]
            status, summary = backend.check(built, rows, on_equal)
            for r, n in enumerate(want_checks):
                got = {built.thr_names[t]: NAMES[int(status[r, t])] for t in range(len(built.thr_names))
                       if status[r, t] != S.NOT_AFFECTED}
                assert got == want_checks[n][k], f"{n} at {now_text}, isThrottledOnEqual={on_equal}: {got}"
