"""Random clusters through two independent routes: manifests -> objects.py -> flat snapshot -> C oracle, and the
manifest-level model of tests/manifest_model.py.  Everything both can say must agree exactly: pod request lists,
`used`, calculated thresholds (and whether they were replaced), throttled flags, next-override instants, and — after the
status has been written back in canonical text — every (pod, throttle) CheckThrottleStatus and PreFilter verdict for
both isThrottledOnEqual values.  This pins the translation layer and the oracle against a restatement that shares
neither data layout nor code with them (only the Quantity / RFC3339 parsers, which have their own cross-checks)."""
import random
from fractions import Fraction

import numpy as np
import pytest

from kube_throttler_amd import snapshot as S
from kube_throttler_amd.objects import ClusterState
from kube_throttler_amd.quantity import parse_rfc3339
from manifest_model import Model, pod_request_resource_list

NOW_TEXT = "2026-01-01T00:00:00Z"
LATER_TEXT = "2026-01-20T00:00:00Z"
TIMES = ["", "", "2025-12-01T00:00:00Z", "2025-12-31T23:59:59Z", "2026-01-01T00:00:00Z", "2026-01-01T00:00:01Z",
         "2026-01-10T00:00:00Z", "2026-02-01T00:00:00+09:00", "not-a-time"]
QTY = {"cpu": ["0", "100m", "250m", "500m", "1"], "memory": ["0", "64Mi", "256Mi", "512Mi"], "amd.com/gpu": ["0", "1", "2"]}
THR_QTY = {"cpu": ["500m", "1", "2", "4"], "memory": ["512Mi", "1Gi", "2Gi", "1536Mi"], "amd.com/gpu": ["1", "2", "4"]}


def _labels(r, keys=("app", "tier", "team"), vals=("x", "y")):
    return {k: r.choice(vals) for k in keys if r.random() < 0.5}


def _requests(r, p=0.6):
    return {name: r.choice(vs) for name, vs in QTY.items() if r.random() < p}


def _selector(r, allow_bad):
    sel = {}
    if r.random() < 0.6:
        sel["matchLabels"] = _labels(r)
    if r.random() < 0.4:
        exprs = []
        for _ in range(r.randint(1, 2)):
            op = r.choice(["In", "NotIn", "Exists", "DoesNotExist"])
            e = {"key": r.choice(["app", "tier", "team"]), "operator": op}
            if op in ("In", "NotIn"):
                e["values"] = r.sample(["x", "y", "z"], r.randint(1, 2))
            exprs.append(e)
        if allow_bad and r.random() < 0.25:
            exprs.append(r.choice([{"key": "app", "operator": "In", "values": []},
                                   {"key": "app", "operator": "Exists", "values": ["x"]},
                                   {"key": "app", "operator": "Bogus"}]))
        sel["matchExpressions"] = exprs
    return sel


def _amount(r, scale=1):
    a = {}
    if r.random() < 0.6:
        a["resourceCounts"] = {"pod": r.randint(0, 4) * scale}
    rr = {}
    for name, vs in THR_QTY.items():
        if r.random() < 0.5:
            rr[name] = "0" if r.random() < 0.08 else r.choice(vs)
    if rr or r.random() < 0.3:
        a["resourceRequests"] = rr
    return a


def random_cluster(seed) -> ClusterState:
    r = random.Random(seed)
    cs = ClusterState()
    namespaces = ["ns0", "ns1", "ns2"]
    for n in namespaces:
        cs.add_namespace(n, {"zone": r.choice(["a", "b"]), "kubernetes.io/metadata.name": n})
    pod_namespaces = namespaces + (["ghost"] if seed % 4 == 0 else [])   # "ghost" has no Namespace object
    for i in range(r.randint(6, 14)):
        spec = {"schedulerName": r.choice(["my-scheduler"] * 4 + ["default-scheduler"]),
                "containers": [{"name": f"c{k}", "resources": {"requests": _requests(r)}} for k in range(r.randint(1, 3))]}
        if r.random() < 0.3:
            spec["initContainers"] = [{"name": "i", "resources": {"requests": _requests(r, 0.5)}}]
        if r.random() < 0.15:
            spec["overhead"] = _requests(r, 0.4)
        phase = "Pending"
        if r.random() < 0.6:
            spec["nodeName"] = "node-1"
            phase = r.choice(["Running", "Running", "Running", "Succeeded", "Failed"])
        cs.add({"kind": "Pod", "metadata": {"name": f"pod{i}", "namespace": r.choice(pod_namespaces), "labels": _labels(r)},
                "spec": spec, "status": {"phase": phase}})
    for i in range(r.randint(3, 7)):
        cluster = r.random() < 0.5
        terms = []
        for _ in range(r.randint(0, 3)):
            t = {"podSelector": _selector(r, allow_bad=seed % 3 == 0)}
            if cluster and r.random() < 0.7:
                t["namespaceSelector"] = (r.choice([{"matchLabels": {"zone": r.choice(["a", "b"])}},
                                                    {"matchExpressions": [{"key": "zone", "operator": "Bogus"}]}])
                                          if r.random() < 0.85 else {})
            terms.append(t)
        spec = {"throttlerName": r.choice(["kube-throttler"] * 5 + ["someone-else"]),
                "selector": {"selectorTerms": terms}, "threshold": _amount(r, 2)}
        if r.random() < 0.5:
            spec["temporaryThresholdOverrides"] = [
                {"begin": r.choice(TIMES), "end": r.choice(TIMES), "threshold": _amount(r, 3)} for _ in range(r.randint(1, 3))]
        md = {"name": f"thr{i}"}
        if not cluster:
            md["namespace"] = r.choice(namespaces)
        cs.add({"kind": "ClusterThrottle" if cluster else "Throttle", "metadata": md, "spec": spec})
        if r.random() < 0.3:
            nn = (md.get("namespace", "") if not cluster else "") + "/" + md["name"]
            cs.reserved[("ClusterThrottle" if cluster else "Throttle", nn)] = {
                "resourceCounts": {"pod": r.randint(1, 2)}, "resourceRequests": {"cpu": r.choice(["100m", "1"])}}
    return cs


def _dense_to_dict(built, v_row, present):
    return {name: Fraction(int(v_row[d])) * Fraction(10) ** built.scales[name]
            for name, d in built.dims.items() if int(present) >> d & 1}


def _compare_reconcile(cs, built, res, rows, model, now, label):
    for k, i in enumerate(rows):
        thr = cs.throttles[i]
        want = model.reconcile(thr, now)
        where = f"{label}/{built.thr_names[i]}"
        assert bool(res.error[k]) == (want is None), f"{where}: reconcile error"
        if want is None:
            continue
        assert built.amount_to_dict(res.used, k) == want["used"].as_dict(), f"{where}: used"
        assert built.amount_to_dict(res.calc, k) == want["calc"].as_dict(), f"{where}: calculated threshold"
        assert bool(res.calc_updated[k]) == want["updated"], f"{where}: calculatedThreshold replaced"
        pod_flag, by_name = want["throttled"]
        assert bool(res.thrl_pod[k]) == pod_flag, f"{where}: throttled.pod"
        got = {name: bool(int(res.thrl_flag[k]) >> d & 1) for name, d in built.dims.items() if int(res.thrl_has[k]) >> d & 1}
        assert got == by_name, f"{where}: throttled.resourceRequests"


@pytest.mark.parametrize("seed", range(160))
def test_random_cluster_both_routes(seed, oracle_mod):
    cs = random_cluster(seed)
    model = Model(cs)
    now, later = parse_rfc3339(NOW_TEXT), parse_rfc3339(LATER_TEXT)
    built = cs.build()
    snap = built.snapshot
    o = oracle_mod.Oracle(snap)
    # ---- resourcelist.PodRequestResourceList
    v, present = o.pod_requests()
    for i, p in enumerate(cs.pods):
        assert _dense_to_dict(built, v[i], present[i]) == pod_request_resource_list(p), f"seed {seed}: requests of pod{i}"
    # ---- reconcile + next override
    need = S.THR_VALID | S.THR_RESPONSIBLE
    rows = np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]
    assert [cs.throttles[i]["spec"]["throttlerName"] == "kube-throttler" for i in range(len(cs.throttles))] == \
        [i in set(rows.tolist()) for i in range(len(cs.throttles))]
    res = o.reconcile(now, rows=rows)
    _compare_reconcile(cs, built, res, rows, model, now, f"seed {seed}")
    nx_s, nx_ns, nx_has = o.next_override(now)
    for k, i in enumerate(rows):
        if res.error[k]:
            continue
        want = model.reconcile(cs.throttles[i], now)["next"]
        got = (int(nx_s[i]), int(nx_ns[i])) if nx_has[i] else None
        assert got == want, f"seed {seed}/{built.thr_names[i]}: next override"
    # ---- UpdateStatus in canonical text, then PreFilter for every pod
    for k, i in enumerate(rows):
        if not res.error[k]:
            full = type("R", (), {})()
            for f in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod"):
                a = np.zeros(snap.n_thr + 1, getattr(res, f).dtype)
                a[i] = getattr(res, f)[k]
                setattr(full, f, a)
            for tab in ("used", "calc"):
                t = S.Amounts(snap.n_thr + 1, snap.D)
                for f in ("v", "present", "count", "has_count"):
                    getattr(t, f)[i] = getattr(getattr(res, tab), f)[k]
                setattr(full, tab, t)
            cs.throttles[i]["status"] = built.status_manifest(full, i, NOW_TEXT, previous=cs.throttles[i].get("status"))
    built = cs.build()
    o = oracle_mod.Oracle(built.snapshot)
    name_of = {S.NOT_THROTTLED: "not-throttled", S.ACTIVE: "active", S.INSUFFICIENT: "insufficient",
               S.EXCEEDS: "pod-requests-exceeds-threshold"}
    for on_equal in (False, True):
        st, sm = o.check(on_equal=on_equal)
        verdict = S.summary_fields(sm)[0]
        for i, p in enumerate(cs.pods):
            want_v, want_st = model.check(p, on_equal)
            where = f"seed {seed}/pod{i} on_equal={on_equal}"
            assert {S.VERDICT_ALLOW: "allow", S.VERDICT_BLOCK: "block", S.VERDICT_ERROR: "error"}[int(verdict[i])] == want_v, where
            if want_v == "error":
                continue
            got = {built.thr_names[t]: name_of[int(st[i, t])] for t in range(len(built.thr_names)) if st[i, t] != S.NOT_AFFECTED}
            assert got == want_st, where
    # ---- a later reconcile against the stored status: replaced only when something changed by value
    rows = np.nonzero((built.snapshot.thr_flags[:built.snapshot.n_thr] & need) == need)[0]
    res = o.reconcile(later, rows=rows)
    _compare_reconcile(cs, built, res, rows, model, later, f"seed {seed} (later)")
