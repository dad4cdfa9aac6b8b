"""Runs the hand-transcribed golden scenarios (tests/golden/*.yaml) against a backend.

A backend exposes ``reconcile(built, now) -> result`` (attributes used, calc, calc_updated, thrl_flag,
thrl_has, thrl_pod, error — rows = all throttles) and ``check(built, rows, on_equal) -> (status, summary)``.
The oracle backend and the GPU engine backend (through the C-ABI) both implement it, so the same
vectors pin the oracle (-m "not gpu") and the HIP path (-m gpu).
"""
from __future__ import annotations

import copy
import json
import os

import yaml

from kube_throttler_amd import snapshot as S
from kube_throttler_amd.objects import ClusterState
from kube_throttler_amd.quantity import format_quantity, parse_quantity, parse_rfc3339, quantity_format

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_STATUS_BY_NAME = {v: k for k, v in S.STATUS_NAMES.items()}
_VERDICT = {"allow": S.VERDICT_ALLOW, "block": S.VERDICT_BLOCK, "error": S.VERDICT_ERROR}


def load_scenarios():
    with open(os.path.join(GOLDEN, "integration_scenarios.yaml")) as fh:
        doc = yaml.safe_load(fh)
    return doc["scenarios"]


def _example():
    with open(os.path.join(GOLDEN, "example_config0.json")) as fh:
        return json.load(fh)["manifests"]


def _pod(name, namespace, labels, cpu=None, scheduled=False, scheduler="my-scheduler", requests=None):
    req = dict(requests or {})
    if cpu is not None:
        req["cpu"] = cpu
    p = {"kind": "Pod", "metadata": {"name": name, "namespace": namespace, "labels": dict(labels or {})},
         "spec": {"schedulerName": scheduler, "containers": [{"name": "ctr", "resources": {"requests": req}}]},
         "status": {"phase": "Pending"}}
    if scheduled:
        p["spec"]["nodeName"] = "node-1"
        p["status"]["phase"] = "Running"
    return p


def _subst(obj, **kw):
    if isinstance(obj, str):
        return obj.format(**kw) if "{" in obj else obj
    if isinstance(obj, dict):
        return {k: _subst(v, **kw) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_subst(v, **kw) for v in obj]
    return obj


def build_cluster_state(sc) -> ClusterState:
    cs = ClusterState(sc["throttlerName"], sc["targetSchedulerName"])
    for n in sc.get("namespaces") or []:
        cs.add_namespace(n["name"], n.get("labels"))
    rep = sc.get("namespaces_repeat")
    if rep:
        for i in range(rep["count"]):
            cs.add_namespace(rep["name"].format(i=i), rep.get("labels"))
    for t in sc.get("throttles") or []:
        cs.add(copy.deepcopy(t))
    rep = sc.get("throttles_repeat")
    if rep:
        for i in range(rep["count"]):
            cs.add(_subst(copy.deepcopy(rep["template"]), i=i))
    if sc.get("example_throttle"):
        t = copy.deepcopy(_example()["throttle"])
        t["metadata"]["namespace"] = "default"
        if sc.get("example_throttle_cpu"):
            t["spec"]["threshold"]["resourceRequests"]["cpu"] = sc["example_throttle_cpu"]
        cs.add(t)
    for name in sc.get("example_pods") or []:
        p = copy.deepcopy(_example()[name])
        p["metadata"]["namespace"] = "default"
        cs.add(p)
    for rep in sc.get("pods_repeat") or []:
        for i in range(rep["count"]):
            cs.add(_pod(rep["name"].format(i=i), rep["namespace"], rep.get("labels"), rep.get("cpu"),
                        rep.get("scheduled", False)))
    grid = sc.get("pods_grid")
    if grid:
        for i in range(grid["namespaces"]):
            for j in range(grid["per_ns"]):
                cs.add(_pod(grid["name"].format(i=i, j=j), grid["ns_name"].format(i=i), grid.get("labels"),
                            grid.get("cpu"), grid.get("scheduled", False)))
    for p in sc.get("pods") or []:
        cs.add(_pod(p["name"], p["namespace"], p.get("labels"), p.get("cpu"), p.get("scheduled", False),
                    requests=p.get("requests")))
    for key, amount in (sc.get("reserved") or {}).items():
        kind, nn = key.split(":", 1)
        cs.reserved[(kind, nn)] = amount
    return cs


def _expected_amount(exp: dict):
    """{pod: n, cpu: "500m", ...} -> (count or None, {name: Fraction})."""
    exp = dict(exp or {})
    count = exp.pop("pod", None)
    return count, {k: parse_quantity(v) for k, v in exp.items()}


def write_back_status(cs: ClusterState, built, res, now_text):
    """UpdateStatus: persist the reconcile result into the manifests (throttle_controller.go:157-173), in the
    canonical text the API server would hold; the next step re-parses it like an informer update would."""
    for i, t in enumerate(cs.throttles):
        if res.error[i]:
            continue
        t["status"] = built.status_manifest(res, i, now_text, previous=t.get("status"))


def check_reconcile_expectation(built, res, i, exp, label):
    count, reqs = _expected_amount(exp.get("used"))
    got = built.amount_to_dict(res.used, i)
    assert got.get("resourceCounts", {}).get("pod") == count, f"{label}: used.pod {got} != {count}"
    assert got.get("resourceRequests", {}) == reqs, f"{label}: used.requests {got} != {reqs}"
    # the expectation is spelled the way the reference's specs spell it (e.g. "1" for 20 x 50m, "200m"): when that
    # spelling is canonical, the written-back text has to be exactly it
    text = built.amount_to_manifest(res.used, i).get("resourceRequests", {})
    for name, want in (exp.get("used") or {}).items():
        if name != "pod" and format_quantity(parse_quantity(want), quantity_format(want)) == want:
            assert text[name] == want, f"{label}: used.{name} written as {text[name]!r}, want {want!r}"
    thr = dict(exp.get("throttled") or {})
    if "pod" in thr:
        assert bool(res.thrl_pod[i]) == thr.pop("pod"), f"{label}: throttled.pod"
    for name, v in thr.items():
        d = built.dims[name]
        assert int(res.thrl_has[i]) >> d & 1, f"{label}: throttled has no key {name}"
        assert bool(int(res.thrl_flag[i]) >> d & 1) == v, f"{label}: throttled.{name}"
    if "calculatedThreshold" in exp:
        count, reqs = _expected_amount(exp["calculatedThreshold"])
        got = built.amount_to_dict(res.calc, i)
        assert got.get("resourceCounts", {}).get("pod") == count, f"{label}: calculatedThreshold.pod"
        assert got.get("resourceRequests", {}) == reqs, f"{label}: calculatedThreshold.requests"


def run_scenario(sc, backend):
    cs = build_cluster_state(sc)
    now = parse_rfc3339(sc["now"])
    for step in sc["steps"]:
        built = cs.build()
        if "schedule" in step:
            for p in cs.pods:
                if p["metadata"]["name"] in step["schedule"]:
                    p["spec"]["nodeName"] = "node-1"
                    p.setdefault("status", {})["phase"] = "Running"
        elif "reconcile" in step or "reconcile_all" in step:
            res = backend.reconcile(built, now)
            if "reconcile_all" in step:
                for i in range(len(cs.throttles)):
                    check_reconcile_expectation(built, res, i, step["reconcile_all"], f"{sc['name']}/{built.thr_names[i]}")
            else:
                for nn, exp in step["reconcile"].items():
                    i = built.thr_names.index(nn)
                    assert not res.error[i]
                    check_reconcile_expectation(built, res, i, exp, f"{sc['name']}/{nn}")
            write_back_status(cs, built, res, sc["now"])
        elif "check" in step:
            names = list(step["check"].keys())
            rows = [next(i for i, p in enumerate(cs.pods) if p["metadata"]["name"] == n) for n in names]
            status, summary = backend.check(built, rows, False)
            verdict, n_exc, n_act, n_ins = S.summary_fields(summary)
            for k, n in enumerate(names):
                exp = step["check"][n]
                label = f"{sc['name']}/{n}"
                want = {built.thr_names.index(nn): _STATUS_BY_NAME[v] for nn, v in exp["statuses"].items()}
                for t in range(len(built.thr_names)):
                    assert int(status[k, t]) == want.get(t, S.NOT_AFFECTED), \
                        f"{label}: status[{built.thr_names[t]}]={S.STATUS_NAMES[int(status[k, t])]}"
                assert int(verdict[k]) == _VERDICT[exp["verdict"]], f"{label}: verdict"
                assert built.reasons(status[k]) == exp["reasons"], f"{label}: reasons {built.reasons(status[k])}"
                exceeded = [r.split("=", 1)[1] for r in exp["reasons"] if "[pod-requests-exceeds-threshold]=" in r]
                ev = built.events(status[k])
                assert len(ev) == (1 if exceeded else 0), f"{label}: events {ev}"
                if exceeded:   # ClusterThrottle names first, then Throttle names — the order of the reason strings
                    assert ev[0]["reason"] == "ResourceRequestsExceedsThrottleThreshold" and \
                        ev[0]["message"].endswith("exceeds their thresholds: " + ",".join(exceeded)), f"{label}: {ev}"
                row = status[k]
                assert (n_exc[k], n_act[k], n_ins[k]) == (int((row == S.EXCEEDS).sum()), int((row == S.ACTIVE).sum()),
                                                          int((row == S.INSUFFICIENT).sum())), f"{label}: summary counts"
        else:
            raise ValueError(step)


class OracleBackend:
    def __init__(self, oracle_mod):
        self.m = oracle_mod

    def reconcile(self, built, now):
        return self.m.Oracle(built.snapshot).reconcile(now)

    def check(self, built, rows, on_equal):
        return self.m.Oracle(built.snapshot).check(rows, on_equal)
