"""Prometheus gauges of kube-throttler, fed straight from the engine's per-throttle vectors.

Same metric families, label sets and units as the reference's recorders
(pkg/controllers/throttle_metrics.go:34-131, clusterthrottle_metrics.go:34-128, metrics_recorder.go:28-66):

    [cluster]throttle_spec_threshold_resourceCounts / _resourceRequests
    [cluster]throttle_status_throttled_resourceCounts / _resourceRequests        (1 = throttled, 0 = not)
    [cluster]throttle_status_used_resourceCounts / _resourceRequests
    [cluster]throttle_status_calculated_threshold_resourceCounts / _resourceRequests

labels: namespace (Throttle only), name, uid, resource.  `cpu` is exported in milli-units (Quantity.MilliValue),
every other resource in units (Quantity.Value) — both round UP like apimachinery does; resourceCounts always carry
resource="pod" (nil counts export 0).
"""
from __future__ import annotations

import math
from fractions import Fraction

from prometheus_client import CollectorRegistry, Gauge, generate_latest

from . import snapshot as S

_FAMILIES = ("spec_threshold", "status_throttled", "status_used", "status_calculated_threshold")
# help texts verbatim (throttle_metrics.go:44-97, clusterthrottle_metrics.go:44-97: both kinds say "of the throttle")
_HELP = {
    ("spec_threshold", "resourceCounts"): "threshold on specific resourceCounts of the throttle",
    ("spec_threshold", "resourceRequests"): "threshold on specific resourceRequests of the throttle",
    ("status_throttled", "resourceCounts"):
        "resourceCounts of the throttle is throttled or not on specific resource (1=throttled, 0=not throttled)",
    ("status_throttled", "resourceRequests"):
        "resourceRequests of the throttle is throttled or not on specific resource (1=throttled, 0=not throttled)",
    ("status_used", "resourceCounts"): "used resource counts of the throttle",
    ("status_used", "resourceRequests"): "used amount of resource requests of the throttle",
    ("status_calculated_threshold", "resourceCounts"): "calculated threshold on specific resourceCounts of the throttle",
    ("status_calculated_threshold", "resourceRequests"): "calculated threshold on specific resourceRequests of the throttle",
}


def _gauge_value(name: str, q: Fraction) -> float:
    """metrics_recorder.go:39-44: MilliValue for cpu, Value otherwise (ceil, as Quantity rounds up)."""
    return float(math.ceil(q * 1000)) if name == "cpu" else float(math.ceil(q))


class MetricsRecorder:
    """Gauges for every throttle of a :class:`kube_throttler_amd.objects.BuiltState`."""

    def __init__(self, registry: CollectorRegistry | None = None):
        self.registry = registry or CollectorRegistry()
        self.g = {}
        for kind, prefix, labels in (("Throttle", "throttle", ["namespace", "name", "uid", "resource"]),
                                     ("ClusterThrottle", "clusterthrottle", ["name", "uid", "resource"])):
            for fam in _FAMILIES:
                for what in ("resourceCounts", "resourceRequests"):
                    self.g[(kind, fam, what)] = Gauge(f"{prefix}_{fam}_{what}", _HELP[(fam, what)], labels,
                                                      registry=self.registry)

    def _labels(self, kind, manifest, resource):
        md = manifest["metadata"]
        lab = {"name": md["name"], "uid": str(md.get("uid", "")), "resource": resource}
        if kind == "Throttle":
            lab["namespace"] = md.get("namespace", "default")
        return lab

    def _amount(self, built, kind, manifest, fam, amounts, i):
        gc, gr = self.g[(kind, fam, "resourceCounts")], self.g[(kind, fam, "resourceRequests")]
        gc.labels(**self._labels(kind, manifest, "pod")).set(float(amounts.count[i]) if amounts.has_count[i] else 0.0)
        for name, q in (built.amount_to_dict(amounts, i).get("resourceRequests") or {}).items():
            gr.labels(**self._labels(kind, manifest, name)).set(_gauge_value(name, q))

    def record(self, built, reconcile=None):
        """Record spec + status of every throttle.  `reconcile`: an engine ReconcileResult (rows = throttle rows) whose
        used / calculated threshold / throttled flags replace the snapshot's stored status (what UpdateStatus persists)."""
        snap = built.snapshot
        used = reconcile.used if reconcile is not None else snap.thr_used
        calc = reconcile.calc if reconcile is not None else snap.thr_calc
        for i, manifest in enumerate(built.cs.throttles):
            # the reference records inside reconcile (throttle_controller.go:159,187), which only ever runs for throttles
            # this throttler is responsible for (event handlers, :404-420)
            if not int(snap.thr_flags[i]) & S.THR_RESPONSIBLE:
                continue
            kind = manifest["kind"]
            self._amount(built, kind, manifest, "spec_threshold", snap.thr_spec, i)
            self._amount(built, kind, manifest, "status_used", used, i)
            self._amount(built, kind, manifest, "status_calculated_threshold", calc, i)
            if reconcile is not None:
                pod, flag, has = bool(reconcile.thrl_pod[i]), int(reconcile.thrl_flag[i]), int(reconcile.thrl_has[i])
            else:
                pod = bool(int(snap.thr_flags[i]) & S.THR_THROTTLED_POD)
                flag, has = int(snap.thr_thrl_flag[i]), int(snap.thr_thrl_has[i])
            self.g[(kind, "status_throttled", "resourceCounts")].labels(**self._labels(kind, manifest, "pod")).set(float(pod))
            for name, d in built.dims.items():
                if has >> d & 1:  # recordIsResourceRequestsThrottled: one sample per key of the map
                    self.g[(kind, "status_throttled", "resourceRequests")].labels(
                        **self._labels(kind, manifest, name)).set(float(flag >> d & 1))
        return self

    def exposition(self) -> str:
        return generate_latest(self.registry).decode()
