"""Host-side object model: Kubernetes-shaped manifests (dicts) -> flat :class:`Snapshot`.

This is the translation the Go shim performs in front of the C-ABI: intern label
strings, resolve resource names to dimensions, turn ``resource.Quantity`` text into exact integers at
a per-dimension scale, parse override instants and validate label selectors.  Field names follow the
reference CRD types (pkg/apis/schedule/v1alpha1/throttle_types.go, clusterthrottle_types.go,
throttle_selector.go, clusterthrottle_selector.go, temporary_threshold_override.go) and core/v1 Pod.
"""
from __future__ import annotations

import hashlib
import re
from fractions import Fraction

import numpy as np

from . import snapshot as S
from .quantity import (DECIMAL_SI, format_quantity, min_scale, parse_quantity, parse_rfc3339, quantity_format,
                       to_scaled)

_QNAME = re.compile(r"^([A-Za-z0-9]([-A-Za-z0-9_.]*[A-Za-z0-9])?)$")
_DNS1123_SUB = re.compile(r"^[a-z0-9]([-a-z0-9]*[a-z0-9])?(\.[a-z0-9]([-a-z0-9]*[a-z0-9])?)*$")
_OPS = {"In": S.OP_IN, "NotIn": S.OP_NOT_IN, "Exists": S.OP_EXISTS, "DoesNotExist": S.OP_DOES_NOT_EXIST}


def _valid_label_key(k: str) -> bool:
    parts = k.split("/")
    if len(parts) == 1:
        name, prefix = parts[0], None
    elif len(parts) == 2:
        prefix, name = parts
        if not prefix or len(prefix) > 253 or not _DNS1123_SUB.match(prefix):
            return False
    else:
        return False
    return 0 < len(name) <= 63 and bool(_QNAME.match(name))


def _valid_label_value(v: str) -> bool:
    return len(v) <= 63 and (v == "" or bool(_QNAME.match(v)))


def selector_requirements(sel: dict | None):
    """metav1.LabelSelectorAsSelector restated: -> (requirements [(op, key, [values])], invalid: bool).

    matchLabels k=v becomes In{k,[v]}; unknown operator, In/NotIn without values, Exists/DoesNotExist
    with values, or malformed key/value make the selector invalid."""
    reqs, invalid = [], False
    sel = sel or {}
    for k, v in sorted((sel.get("matchLabels") or {}).items()):
        if not _valid_label_key(k) or not _valid_label_value(str(v)):
            invalid = True
        reqs.append((S.OP_IN, k, [str(v)]))
    for e in sel.get("matchExpressions") or []:
        op = _OPS.get(e.get("operator"))
        vals = [str(x) for x in (e.get("values") or [])]
        key = e.get("key", "")
        if op is None:
            invalid = True
            continue
        if op in (S.OP_IN, S.OP_NOT_IN) and not vals:
            invalid = True
        if op in (S.OP_EXISTS, S.OP_DOES_NOT_EXIST) and vals:
            invalid = True
        if not _valid_label_key(key) or not all(_valid_label_value(x) for x in vals):
            invalid = True
        reqs.append((op, key, vals))
    return reqs, invalid


def messages_fingerprint(msgs) -> int:
    if not msgs:
        return 0
    h = hashlib.blake2b("\x00".join(msgs).encode(), digest_size=8).digest()
    return int.from_bytes(h, "little") or 1


class Interner:
    def __init__(self):
        self.ids: dict = {}

    def __call__(self, x) -> int:
        i = self.ids.get(x)
        if i is None:
            i = self.ids[x] = len(self.ids) + 1
        return i


class ClusterState:
    """A bag of manifests plus the per-throttle reserved amounts of the scheduler-side cache."""

    def __init__(self, throttler_name="kube-throttler", target_scheduler_name="my-scheduler"):
        self.throttler_name = throttler_name
        self.target_scheduler_name = target_scheduler_name
        self.namespaces: list[dict] = []
        self.pods: list[dict] = []
        self.throttles: list[dict] = []  # kind Throttle / ClusterThrottle
        self.reserved: dict = {}  # (kind, "ns/name") -> {"resourceCounts":..., "resourceRequests":...}
        self.missing_namespaces: set = set()  # namespaces referenced by pods but without an object

    # ---- convenience constructors (mirror the reference test helpers mkPod/mkNamespace) -------
    def add_namespace(self, name, labels=None):
        self.namespaces.append({"metadata": {"name": name, "labels": dict(labels or {})}})
        return self

    def add(self, manifest: dict):
        kind = manifest.get("kind")
        if kind == "Pod":
            self.pods.append(manifest)
        elif kind in ("Throttle", "ClusterThrottle"):
            self.throttles.append(manifest)
        elif kind == "Namespace":
            self.namespaces.append(manifest)
        else:
            raise ValueError(f"unsupported kind {kind!r}")
        return manifest

    # ---- flattening ---------------------------------------------------------------------------
    def build(self, only=None) -> "BuiltState":
        """``only``: keep just these resource names as dimensions (a page of a cluster with more than KT_MAX_DIMS of
        them, see build_pages); every other name is ignored wherever it occurs."""
        return BuiltState(self, only)

    def resource_names(self) -> list:
        """Every resource name that occurs in a request, an overhead, a threshold, an override, a status or a reservation."""
        names = set()

        def see(rl):
            names.update((rl or {}).keys())

        for p in self.pods:
            spec = p.get("spec", {})
            for c in (spec.get("containers") or []) + (spec.get("initContainers") or []):
                see((c.get("resources") or {}).get("requests"))
            see(spec.get("overhead"))
        for t in self.throttles:
            spec, st = t.get("spec", {}), t.get("status", {}) or {}
            see((spec.get("threshold") or {}).get("resourceRequests"))
            for o in spec.get("temporaryThresholdOverrides") or []:
                see((o.get("threshold") or {}).get("resourceRequests"))
            see(((st.get("calculatedThreshold") or {}).get("threshold") or {}).get("resourceRequests"))
            see((st.get("used") or {}).get("resourceRequests"))
            see((st.get("throttled") or {}).get("resourceRequests"))
        for a in self.reserved.values():
            see((a or {}).get("resourceRequests"))
        return sorted(names)

    def build_pages(self, max_dims: int = S.KT_MAX_DIMS) -> list:
        """The reference sums ANY resource name (pkg/resourcelist/resourcelist.go:27-54); an engine holds KT_MAX_DIMS
        dimensions.  A cluster with more names is built as several PAGES — the same pods, namespaces, throttles and
        selectors, each page with its own <= max_dims names as dimensions — evaluated by one engine each and combined by
        kube_throttler_amd/paging.py (every step of CheckThrottledFor / IsThrottled is an OR over resource names plus a
        count part that every page computes alike, so the combination is exact)."""
        names = self.resource_names()
        if len(names) <= max_dims:
            return [self.build()]
        return [self.build(only=set(names[i:i + max_dims])) for i in range(0, len(names), max_dims)]


def _amount_quantities(a: dict | None):
    return [parse_quantity(q) for q in ((a or {}).get("resourceRequests") or {}).values()]


class BuiltState:
    """Flattened state: ``.snapshot`` plus the dictionaries needed to read results back."""

    def __init__(self, cs: ClusterState, only=None):
        self.cs = cs
        self.only = only  # None, or the resource names this page keeps
        ns_names = [n["metadata"]["name"] for n in cs.namespaces]
        for p in cs.pods:
            ns = p["metadata"].get("namespace", "default")
            if ns not in ns_names:
                ns_names.append(ns)
                cs.missing_namespaces.add(ns)
        for t in cs.throttles:
            if t["kind"] == "Throttle":
                ns = t["metadata"].get("namespace", "default")
                if ns not in ns_names:
                    ns_names.append(ns)
                    cs.missing_namespaces.add(ns)
        self.ns_index = {n: i for i, n in enumerate(ns_names)}
        ns_objs = {n["metadata"]["name"]: n for n in cs.namespaces}

        # ---- resource dimensions and scales
        quantities: dict[str, list[Fraction]] = {}

        # Format (suffix family) of the first non-zero quantity seen per resource name: what a `used` sum inherits
        # through Quantity.Add when a resource is always written in one family (status write-back only)
        self.formats: dict[str, str] = {}

        def see(rl):
            for name, q in (rl or {}).items():
                if only is not None and name not in only:
                    continue
                v = parse_quantity(q)
                quantities.setdefault(name, []).append(v)
                if v != 0:
                    self.formats.setdefault(name, quantity_format(q))

        for p in cs.pods:
            spec = p.get("spec", {})
            for c in (spec.get("containers") or []) + (spec.get("initContainers") or []):
                see((c.get("resources") or {}).get("requests"))
            see(spec.get("overhead"))
        for t in cs.throttles:
            spec, st = t.get("spec", {}), t.get("status", {}) or {}
            see((spec.get("threshold") or {}).get("resourceRequests"))
            for o in spec.get("temporaryThresholdOverrides") or []:
                see((o.get("threshold") or {}).get("resourceRequests"))
            see(((st.get("calculatedThreshold") or {}).get("threshold") or {}).get("resourceRequests"))
            see((st.get("used") or {}).get("resourceRequests"))
            for name in ((st.get("throttled") or {}).get("resourceRequests") or {}):
                if only is None or name in only:
                    quantities.setdefault(name, [])
        for a in cs.reserved.values():
            see((a or {}).get("resourceRequests"))
        self.dims = {name: i for i, name in enumerate(sorted(quantities))}
        self.scales = {name: min_scale(vs) for name, vs in quantities.items()}
        D = max(len(self.dims), 1)
        if D > S.KT_MAX_DIMS:
            raise ValueError("too many resource dimensions")

        self.key_id, self.pair_id = Interner(), Interner()

        def labels_of(d):
            return sorted(((d or {}).items()))

        max_l = 0
        for n in cs.namespaces:
            max_l = max(max_l, len(n["metadata"].get("labels") or {}))
        for p in cs.pods:
            max_l = max(max_l, len(p["metadata"].get("labels") or {}))
        snap = S.Snapshot(D, min(max(max_l, 1), S.KT_MAX_LABELS))
        self.snapshot = snap

        # ---- namespaces
        n_ns = len(ns_names)
        tot = sum(len(ns_objs.get(n, {}).get("metadata", {}).get("labels") or {}) for n in ns_names)
        snap.alloc_namespaces(n_ns, tot)
        k = 0
        for i, n in enumerate(ns_names):
            snap.ns_valid[i] = 1 if n in ns_objs else 0
            for key, val in labels_of(ns_objs.get(n, {}).get("metadata", {}).get("labels")):
                snap.ns_label_key[k] = self.key_id(key)
                snap.ns_label_pair[k] = self.pair_id((key, str(val)))
                k += 1
            snap.ns_label_off[i + 1] = k

        # ---- pods
        n_lab = sum(len(p["metadata"].get("labels") or {}) for p in cs.pods)
        n_ctr = sum(len(p.get("spec", {}).get("containers") or []) + len(p.get("spec", {}).get("initContainers") or [])
                    for p in cs.pods)
        snap.alloc_pods(len(cs.pods), n_lab, n_ctr)
        self.pod_names = []
        k = c_i = 0
        for i, p in enumerate(cs.pods):
            md, spec, st = p["metadata"], p.get("spec", {}), p.get("status", {}) or {}
            ns = md.get("namespace", "default")
            self.pod_names.append(f"{ns}/{md['name']}")
            snap.pod_ns[i] = self.ns_index[ns]
            f = S.POD_VALID
            if spec.get("schedulerName", "default-scheduler") == cs.target_scheduler_name:
                f |= S.POD_SCHED_MATCH
            if spec.get("nodeName"):
                f |= S.POD_SCHEDULED
            if st.get("phase") in ("Succeeded", "Failed"):
                f |= S.POD_FINISHED
            snap.pod_flags[i] = f
            for key, val in labels_of(md.get("labels")):
                snap.pod_label_key[k] = self.key_id(key)
                snap.pod_label_pair[k] = self.pair_id((key, str(val)))
                k += 1
            snap.pod_label_off[i + 1] = k
            for init, cl in ((1, spec.get("initContainers") or []), (0, spec.get("containers") or [])):
                for c in cl:
                    snap.ctr_init[c_i] = init
                    self._fill_row(snap.ctr_req, snap.ctr_present, c_i, (c.get("resources") or {}).get("requests"))
                    c_i += 1
            snap.pod_ctr_off[i + 1] = c_i
            if spec.get("overhead") is not None:
                self._fill_row(snap.pod_ovh, snap.pod_ovh_present, i, spec["overhead"])
                snap.pod_ovh_present[i] |= np.uint32(1 << 31)

        # ---- throttles
        n_ovr = sum(len(t.get("spec", {}).get("temporaryThresholdOverrides") or []) for t in cs.throttles)
        n_term = sum(len((t.get("spec", {}).get("selector") or {}).get("selectorTerms") or []) for t in cs.throttles)
        snap.alloc_throttles(len(cs.throttles), n_ovr, n_term)
        self.thr_names, self.thr_kinds, self.thr_messages = [], [], []
        o_i = term_i = 0
        for i, t in enumerate(cs.throttles):
            md, spec, st = t["metadata"], t.get("spec", {}), t.get("status", {}) or {}
            cluster = t["kind"] == "ClusterThrottle"
            ns = "" if cluster else md.get("namespace", "default")
            # types.NamespacedName.String(): ClusterThrottle renders as "/name" (plugin.go:289-295)
            self.thr_names.append(f"{ns}/{md['name']}")
            self.thr_kinds.append(t["kind"])
            f = S.THR_VALID | (S.THR_CLUSTER if cluster else 0)
            if spec.get("throttlerName") == cs.throttler_name:
                f |= S.THR_RESPONSIBLE
            snap.thr_ns[i] = 0 if cluster else self.ns_index[ns]
            self._fill_amount(snap.thr_spec, i, spec.get("threshold"))
            ct = st.get("calculatedThreshold") or {}
            self._fill_amount(snap.thr_calc, i, ct.get("threshold"))
            if ct.get("calculatedAt"):
                f |= S.THR_CALC_AT_NONZERO
            self._fill_amount(snap.thr_used, i, st.get("used"))
            thrl = st.get("throttled") or {}
            if (thrl.get("resourceCounts") or {}).get("pod"):
                f |= S.THR_THROTTLED_POD
            for name, v in (thrl.get("resourceRequests") or {}).items():
                if name not in self.dims:
                    continue  # another page's resource
                snap.thr_thrl_has[i] |= np.uint32(1 << self.dims[name])
                if v:
                    snap.thr_thrl_flag[i] |= np.uint32(1 << self.dims[name])
            key = (t["kind"], self.thr_names[-1])
            self._fill_amount(snap.thr_reserved, i, cs.reserved.get(key))
            snap.thr_status_msgs_fp[i] = messages_fingerprint(ct.get("messages"))
            msgs = []
            for j, o in enumerate(spec.get("temporaryThresholdOverrides") or []):
                err = None
                for field, label in (("begin", "Begin"), ("end", "End")):
                    txt = o.get(field) or ""
                    if txt == "":
                        sec, nsec = S.ZERO_TIME_S, 0
                    else:
                        try:
                            sec, nsec = parse_rfc3339(txt)
                        except ValueError as e:
                            err = err or f"index {j}: Failed to parse {label}: {e}"
                            sec, nsec = S.ZERO_TIME_S, 0
                    if field == "begin":
                        snap.ovr_begin_s[o_i], snap.ovr_begin_ns[o_i] = sec, nsec
                    else:
                        snap.ovr_end_s[o_i], snap.ovr_end_ns[o_i] = sec, nsec
                if err:
                    snap.ovr_flags[o_i] |= S.OVR_PARSE_ERROR
                    if "Failed to parse End" in err:
                        snap.ovr_flags[o_i] |= S.OVR_BEGIN_PARSED
                    msgs.append(err)
                self._fill_amount(snap.ovr_thr, o_i, o.get("threshold"))
                o_i += 1
            snap.thr_ovr_off[i + 1] = o_i
            self.thr_messages.append(msgs)
            snap.thr_spec_msgs_fp[i] = messages_fingerprint(msgs)
            for term in (spec.get("selector") or {}).get("selectorTerms") or []:
                reqs, invalid = selector_requirements(term.get("podSelector"))
                if invalid:
                    snap.term_flags[term_i] |= S.TERM_POD_SEL_INVALID
                for op, k_, vals in reqs:
                    snap.preq.add(op, self.key_id(k_), [self.pair_id((k_, v)) for v in vals])
                snap.term_preq_off[term_i + 1] = len(snap.preq)
                if cluster:
                    reqs, invalid = selector_requirements(term.get("namespaceSelector"))
                    if invalid:
                        snap.term_flags[term_i] |= S.TERM_NS_SEL_INVALID
                    for op, k_, vals in reqs:
                        snap.nreq.add(op, self.key_id(k_), [self.pair_id((k_, v)) for v in vals])
                snap.term_nreq_off[term_i + 1] = len(snap.nreq)
                term_i += 1
            snap.thr_term_off[i + 1] = term_i
            snap.thr_flags[i] = f

    # ---- helpers --------------------------------------------------------------------------------
    def _fill_row(self, v, present, i, rl):
        for name, q in (rl or {}).items():
            if name not in self.dims:
                continue  # another page's resource (ClusterState.build_pages)
            d = self.dims[name]
            v[i, d] = to_scaled(parse_quantity(q), self.scales[name])
            present[i] |= np.uint32(1 << d)

    def _fill_amount(self, amounts: S.Amounts, i, a):
        a = a or {}
        rc = a.get("resourceCounts")
        amounts.has_count[i] = 1 if rc is not None else 0
        amounts.count[i] = int(rc.get("pod", 0)) if rc is not None else 0
        self._fill_row(amounts.v, amounts.present, i, a.get("resourceRequests"))

    def amount_to_dict(self, amounts: S.Amounts, i) -> dict:
        """Dense row -> {"resourceCounts": {"pod": n}?, "resourceRequests": {name: Fraction}}."""
        out = {}
        if amounts.has_count[i]:
            out["resourceCounts"] = {"pod": int(amounts.count[i])}
        rr = {}
        for name, d in self.dims.items():
            if int(amounts.present[i]) >> d & 1:
                rr[name] = Fraction(int(amounts.v[i, d])) * Fraction(10) ** self.scales[name]
        if rr:
            out["resourceRequests"] = rr
        return out

    def amount_to_manifest(self, amounts: S.Amounts, i) -> dict:
        """Dense row -> the ResourceAmount as the API server persists it: ``Quantity.String()`` per resource."""
        out = self.amount_to_dict(amounts, i)
        if "resourceRequests" in out:
            out["resourceRequests"] = {k: format_quantity(v, self.formats.get(k, DECIMAL_SI))
                                       for k, v in out["resourceRequests"].items()}
        return out

    def status_manifest(self, res, i, now_text: str, previous: dict | None = None) -> dict:
        """``status`` of throttle row ``i`` after a reconcile, as UpdateStatus would write it
        (throttle_controller.go:116-133,157-175): ``used`` always replaced, ``calculatedThreshold`` only when the
        engine reports it replaced (threshold or messages changed by value: ``calculatedAt`` := now), ``throttled``
        with one key per calculated-threshold resource.  ``res`` carries used, calc, calc_updated, thrl_flag, thrl_has,
        thrl_pod (rows = all throttles)."""
        st = dict(previous or {})
        st["used"] = self.amount_to_manifest(res.used, i)
        if res.calc_updated[i]:
            st["calculatedThreshold"] = {"threshold": self.amount_to_manifest(res.calc, i), "calculatedAt": now_text,
                                         "messages": list(self.thr_messages[i])}
        else:
            st.setdefault("calculatedThreshold", {})
        thr = {"resourceCounts": {"pod": bool(res.thrl_pod[i])}, "resourceRequests": {}}
        for name, d in self.dims.items():
            if int(res.thrl_has[i]) >> d & 1:
                thr["resourceRequests"][name] = bool(int(res.thrl_flag[i]) >> d & 1)
        st["throttled"] = thr
        return st

    def events(self, status_row) -> list[dict]:
        """The Warning event PreFilter records when the pod's own requests exceed a threshold (plugin.go:190-202):
        ClusterThrottle names first, then Throttle names."""
        names = [self.thr_names[t] for kind in ("ClusterThrottle", "Throttle") for t in range(len(self.thr_names))
                 if self.thr_kinds[t] == kind and status_row[t] == S.EXCEEDS]
        if not names:
            return []
        return [{"type": "Warning", "reason": "ResourceRequestsExceedsThrottleThreshold",
                 "message": "It won't be scheduled unless decreasing resource requests or increasing ClusterThrottle/Throttle "
                            "threshold because its resource requests exceeds their thresholds: " + ",".join(names)}]

    def reasons(self, status_row) -> list[str]:
        """PreFilter reason strings in the reference's fixed order (plugin.go:182-214)."""
        out = []
        for code, label in ((S.EXCEEDS, "pod-requests-exceeds-threshold"), (S.ACTIVE, "active"),
                            (S.INSUFFICIENT, "insufficient")):
            for kind, tag in (("ClusterThrottle", "clusterthrottle"), ("Throttle", "throttle")):
                names = [self.thr_names[t] for t in range(len(self.thr_names))
                         if self.thr_kinds[t] == kind and status_row[t] == code]
                if names:
                    out.append(f"{tag}[{label}]={','.join(names)}")
        return out
