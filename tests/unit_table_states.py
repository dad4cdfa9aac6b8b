"""Collects every cluster state the reference's unit-test tables build (tests/test_oracle_unit_tables.py) so that the
same states can be replayed end to end: pod requests, reconcile, next override, check with both isThrottledOnEqual
values, at the instants the tables probe.  The CPU suite replays them on the oracle alone (sanity of the replay
itself), the GPU suite through the C-ABI against the oracle (tests/test_parity_extended_gpu.py)."""
import copy
import inspect

import test_oracle_unit_tables as U

from kube_throttler_amd.quantity import parse_rfc3339

NOW_2026 = (1767225600, 0)


def collect(oracle_mod):
    """-> list of (label, ClusterState).  Every throttle is made one this throttler is responsible for, so the replay
    reconciles it (the tables use foreign throttler names where they only need the object as a container)."""
    U.RECORDED = []
    marks = []
    try:
        for name, fn in sorted(inspect.getmembers(U, inspect.isfunction)):
            if not name.startswith("test_"):
                continue
            before = len(U.RECORDED)
            if "b" in inspect.signature(fn).parameters:
                fn(oracle_mod, False)
            else:
                fn(oracle_mod)
            marks += [f"{name[5:]}#{k}" for k in range(len(U.RECORDED) - before)]
        states = [copy.deepcopy(cs) for cs in U.RECORDED]
    finally:
        U.RECORDED = None
    out = []
    for label, cs in zip(marks, states):
        for t in cs.throttles:
            t["spec"]["throttlerName"] = cs.throttler_name
        if not cs.pods:   # the override tables build no pod: give the check something to look at
            ns = cs.namespaces[0]["metadata"]["name"]
            cs.add(U.mk_pod("probe", ns, {"probe": "probe"}))
        out.append((label, cs))
        if cs.pods:   # the same state once its pods are bound: they now count into `used`
            bound = copy.deepcopy(cs)
            for p in bound.pods:
                p["spec"]["nodeName"] = "node-1"
                p.setdefault("status", {})["phase"] = "Running"
            out.append((label + "/bound", bound))
    return out


def instants(cs):
    """The instants worth probing: a fixed one, plus every override boundary of the state -1 s / +0 / +1 s."""
    out = [NOW_2026]
    for t in cs.throttles:
        for o in t["spec"].get("temporaryThresholdOverrides") or []:
            for key in ("begin", "end"):
                try:
                    s, ns = parse_rfc3339(o.get(key) or "")
                except ValueError:
                    continue
                out += [(s - 1, ns), (s, ns), (s + 1, ns)]
    seen, uniq = set(), []
    for x in out:
        if x not in seen:
            seen.add(x)
            uniq.append(x)
    return uniq
