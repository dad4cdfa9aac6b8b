"""A second, independent restatement of the hot path — at the MANIFEST level (strings, dicts, exact Fractions), object by
object like the Go code, with none of the engine's vocabulary (no dimensions, scales, ids, masks).  It exists to pin the
translation layer (kube_throttler_amd/objects.py: manifests -> flat snapshot) together with the C oracle: random
clusters must come out identical on both routes (tests/test_manifest_model.py).  Test infrastructure only.

Every function cites the reference lines it follows (paths under /root/reference).
"""
from __future__ import annotations

from fractions import Fraction

from kube_throttler_amd.quantity import parse_quantity, parse_rfc3339

ZERO_TIME = (-62135596800, 0)   # Go's time.Time{} as (unix seconds, nanoseconds)


class SelectorError(Exception):
    pass


# ---------------------------------------------------------------- k8s.io/apimachinery LabelSelectorAsSelector
def selector_matches(sel: dict | None, labels: dict) -> bool:
    """metav1.LabelSelectorAsSelector(sel).Matches(labels); raises SelectorError where the conversion fails."""
    sel = sel or {}
    reqs = [(k, "In", [v]) for k, v in (sel.get("matchLabels") or {}).items()]
    for e in sel.get("matchExpressions") or []:
        reqs.append((e["key"], e["operator"], list(e.get("values") or [])))
    for key, op, values in reqs:                       # the whole selector converts before anything is matched
        if op in ("In", "NotIn"):
            if not values:
                raise SelectorError(f"{op}: values must be non-empty")
        elif op in ("Exists", "DoesNotExist"):
            if values:
                raise SelectorError(f"{op}: values must be empty")
        else:
            raise SelectorError(f"{op!r} is not a valid label selector operator")
    for key, op, values in reqs:
        if op == "In" and not (key in labels and labels[key] in values):
            return False
        if op == "NotIn" and key in labels and labels[key] in values:
            return False
        if op == "Exists" and key not in labels:
            return False
        if op == "DoesNotExist" and key in labels:
            return False
    return True


# ---------------------------------------------------------------- pkg/resourcelist/resourcelist.go
def _rl(d: dict | None) -> dict:
    return {k: parse_quantity(v) for k, v in (d or {}).items()}


def rl_add(lhs: dict, rhs: dict):                      # :48-54 (the key is created even when 0 is added)
    for name, q in rhs.items():
        lhs[name] = lhs.get(name, Fraction(0)) + q


def rl_set_max(lhs: dict, rhs: dict):                  # :76-84 (a key only rhs has is copied, zero or not)
    for name, q in rhs.items():
        lhs[name] = max(lhs[name], q) if name in lhs else q


def pod_request_resource_list(pod: dict) -> dict:      # :27-46
    spec = pod.get("spec") or {}
    ic = {}
    for c in spec.get("initContainers") or []:
        rl_set_max(ic, _rl((c.get("resources") or {}).get("requests")))
    cs = {}
    for c in spec.get("containers") or []:
        rl_add(cs, _rl((c.get("resources") or {}).get("requests")))
    rl_set_max(cs, ic)
    if spec.get("overhead") is not None:
        rl_add(cs, _rl(spec["overhead"]))
    return cs


# ---------------------------------------------------------------- pkg/apis/schedule/v1alpha1/resource_amount.go
class Amount:
    """ResourceAmount: counts is None for a nil resourceCounts."""

    def __init__(self, counts=None, requests=None):
        self.counts, self.requests = counts, dict(requests or {})

    @classmethod
    def of_manifest(cls, a: dict | None):
        a = a or {}
        rc = a.get("resourceCounts")
        return cls(None if rc is None else int(rc.get("pod", 0)), _rl(a.get("resourceRequests")))

    def add(self, b: "Amount") -> "Amount":           # :91-110
        out = Amount(self.counts, self.requests)
        if out.counts is None:
            out.counts = b.counts
        elif b.counts is not None:
            out.counts += b.counts
        rl_add(out.requests, b.requests)
        return out

    def as_dict(self) -> dict:
        out = {}
        if self.counts is not None:
            out["resourceCounts"] = {"pod": self.counts}
        if self.requests:
            out["resourceRequests"] = dict(self.requests)
        return out


def amount_of_pod(pod: dict) -> Amount:                # :71-76
    return Amount(1, pod_request_resource_list(pod))


def is_throttled(threshold: Amount, used: Amount, on_equal: bool):   # :127-159 -> (pod flag, {name: flag})
    def over(u, t):
        return u >= t if on_equal else u > t
    pod = threshold.counts is not None and used.counts is not None and over(used.counts, threshold.counts)
    return pod, {rn: (rn in used.requests and over(used.requests[rn], qt)) for rn, qt in threshold.requests.items()}


def is_throttled_for(flags, pod: dict) -> bool:        # :46-65
    pod_flag, by_name = flags
    if pod_flag:
        return True
    for rn, rq in amount_of_pod(pod).requests.items():
        if rq == 0:
            continue
        if by_name.get(rn, False):
            return True
    return False


# ---------------------------------------------------------------- temporary_threshold_override.go, throttle_types.go
def _instant(text):                                    # temporary_threshold_override.go:33-55
    return ZERO_TIME if not text else parse_rfc3339(text)


def override_is_active(o: dict, now) -> bool:          # :57-70 ; raises ValueError on unparsable begin/end
    begin, end = _instant(o.get("begin")), _instant(o.get("end"))
    return begin <= now and (end == ZERO_TIME or now <= end)


def calculate_threshold(spec: dict, now):              # throttle_types.go:65-106 -> (Amount, [index of errored overrides])
    found, merged, errored = False, Amount(None, {}), []
    for i, o in enumerate(spec.get("temporaryThresholdOverrides") or []):
        try:
            active = override_is_active(o, now)
        except ValueError:
            errored.append(i)
            continue
        if active:
            found = True
            thr = Amount.of_manifest(o.get("threshold"))
            if merged.counts is None and thr.counts is not None:
                merged.counts = thr.counts
            for rn, rq in thr.requests.items():
                merged.requests.setdefault(rn, rq)
    return (merged if found else Amount.of_manifest(spec.get("threshold"))), errored


def next_override(spec: dict, now):                    # throttle_types.go:37-63 -> instant or None
    best = None
    for o in spec.get("temporaryThresholdOverrides") or []:
        try:
            begin = _instant(o.get("begin"))
        except ValueError:
            continue
        if begin > now:
            best = begin if best is None or begin < best else best
        try:
            end = _instant(o.get("end"))
        except ValueError:
            continue
        if end > now:
            best = end if best is None or end < best else best
    return best


def check_throttled_for(kind: str, thr: dict, pod: dict, reserved: Amount, on_equal: bool) -> str:
    """throttle_types.go:128-153 / clusterthrottle_types.go:30-55."""
    st = thr.get("status") or {}
    ct = st.get("calculatedThreshold") or {}
    threshold = Amount.of_manifest((thr.get("spec") or {}).get("threshold"))
    if ct.get("calculatedAt"):
        threshold = Amount.of_manifest(ct.get("threshold"))
    if is_throttled_for(is_throttled(threshold, amount_of_pod(pod), False), pod):
        return "pod-requests-exceeds-threshold"
    stored = st.get("throttled") or {}
    if is_throttled_for((bool((stored.get("resourceCounts") or {}).get("pod", False)),
                         dict(stored.get("resourceRequests") or {})), pod):
        return "active"
    used = Amount.of_manifest(st.get("used"))
    already = Amount().add(used).add(reserved)
    if is_throttled_for(is_throttled(threshold, already, True if kind == "Throttle" else on_equal), pod):
        return "active"
    total = Amount().add(used).add(amount_of_pod(pod)).add(reserved)
    if is_throttled_for(is_throttled(threshold, total, on_equal), pod):
        return "insufficient"
    return "not-throttled"


# ---------------------------------------------------------------- pkg/controllers/*_controller.go
class Model:
    def __init__(self, cs):
        self.cs = cs
        self.namespaces = {n["metadata"]["name"]: n for n in cs.namespaces}

    # -- selectors (throttle_selector.go:30-54, clusterthrottle_selector.go:30-87)
    def _matches(self, thr, pod) -> bool:
        terms = ((thr.get("spec") or {}).get("selector") or {}).get("selectorTerms") or []
        labels = pod["metadata"].get("labels") or {}
        if thr["kind"] == "Throttle":
            return any(selector_matches(t.get("podSelector"), labels) for t in terms)      # any() stops at a match
        ns = self.namespaces[pod["metadata"]["namespace"]]
        for t in terms:
            if not self._term_matches_namespace(t, ns):
                continue
            if selector_matches(t.get("podSelector"), labels):
                return True
        return False

    @staticmethod
    def _term_matches_namespace(term, ns) -> bool:     # conversion errors are swallowed (:63-69)
        try:
            return selector_matches(term.get("namespaceSelector"), ns["metadata"].get("labels") or {})
        except SelectorError:
            return False

    def _responsible(self, thr) -> bool:               # throttle_controller.go:213-215
        return (thr.get("spec") or {}).get("throttlerName") == self.cs.throttler_name

    def _should_count_in(self, pod) -> bool:           # :217-219, pod_util.go:22-24
        spec = pod.get("spec") or {}
        return spec.get("schedulerName") == self.cs.target_scheduler_name and bool(spec.get("nodeName"))

    # -- reconcile (throttle_controller.go:103-133,221-246 ; clusterthrottle_controller.go:106-136,224-270)
    def reconcile(self, thr, now):
        """-> None for a selector error, else dict(used, calc, updated, throttled=(pod, {name: flag}), next)."""
        if thr["kind"] == "Throttle":
            pods = [p for p in self.cs.pods if p["metadata"]["namespace"] == thr["metadata"]["namespace"]]
        else:
            terms = ((thr.get("spec") or {}).get("selector") or {}).get("selectorTerms") or []
            ok_ns = {name for name, ns in self.namespaces.items() if any(self._term_matches_namespace(t, ns) for t in terms)}
            pods = [p for p in self.cs.pods if p["metadata"]["namespace"] in ok_ns]
        used = Amount()
        try:
            for p in pods:
                if not self._should_count_in(p) or not self._matches(thr, p):
                    continue
                if (p.get("status") or {}).get("phase") not in ("Succeeded", "Failed"):
                    used = used.add(amount_of_pod(p))
        except SelectorError:
            return None
        calc, errored = calculate_threshold(thr.get("spec") or {}, now)
        # the stored calculatedThreshold is replaced only when threshold or messages differ BY VALUE (:120-131)
        stored = ((thr.get("status") or {}).get("calculatedThreshold") or {})
        old = Amount.of_manifest(stored.get("threshold"))
        old_msgs = list(stored.get("messages") or [])
        same_msgs = len(old_msgs) == len(errored) and all(m.startswith(f"index {i}: ") for m, i in zip(old_msgs, errored))
        updated = not (old.counts == calc.counts and old.requests == calc.requests and same_msgs)
        return {"used": used, "calc": calc, "updated": updated, "throttled": is_throttled(calc, used, True),
                "next": next_override(thr.get("spec") or {}, now)}

    # -- check (plugin.go:148-215 ; throttle_controller.go:248-269,349-397 ; clusterthrottle_controller.go:272-298,378-425)
    def check(self, pod, on_equal=False):
        """-> ("error", {}) or (verdict, {throttle name: status}) with verdict in allow / block."""
        out = {}
        if pod["metadata"]["namespace"] not in self.namespaces:
            return "error", {}                         # namespaceInformer.Lister().Get fails (clusterthrottle_controller.go:273-276)
        try:
            for thr in self.cs.throttles:
                kind = thr["kind"]
                if kind == "Throttle" and thr["metadata"]["namespace"] != pod["metadata"]["namespace"]:
                    continue
                if not self._responsible(thr) or not self._matches(thr, pod):
                    continue
                nn = (thr["metadata"].get("namespace", "") if kind == "Throttle" else "") + "/" + thr["metadata"]["name"]
                reserved = Amount.of_manifest(self.cs.reserved.get((kind, nn)))
                out[nn] = check_throttled_for(kind, thr, pod, reserved, on_equal)
        except SelectorError:
            return "error", {}
        return ("block" if any(s != "not-throttled" for s in out.values()) else "allow"), out
