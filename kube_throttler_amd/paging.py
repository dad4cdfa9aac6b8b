"""More resource names than one engine has dimensions (KT_MAX_DIMS = 16): pages.

The reference sums and compares ANY resource name (pkg/resourcelist/resourcelist.go:27-54, resource_amount.go:127-159).
``ClusterState.build_pages()`` builds the same cluster once per page of <= 16 names; one engine evaluates each page and
this module combines the results.  The combination is exact, not a heuristic: every step of ``CheckThrottledFor``
(throttle_types.go:128-153) is ``count part  OR  exists a resource name ...`` — the count part depends on no resource
name, so every page computes it alike, and the name part of the whole cluster is the OR of the pages' name parts.  Hence
    exceeds      <=> some page says exceeds
    active       <=> no page says exceeds, some page says active
    insufficient <=> no page says exceeds or active, some page says insufficient
and a pod-level Error (selector / namespace) shows in every page.  ``used``, ``calculatedThreshold`` and ``throttled`` of
a reconcile are per resource name — each name comes from the page that owns it — while pod counts, the pod flag and the
next-override instant are the same in every page; ``calculatedThreshold`` counts as replaced when any page replaced it.
"""
from __future__ import annotations

import numpy as np

from . import engine as E
from . import snapshot as S

# CheckThrottleStatus precedence of throttle_types.go:128-153 (first hit wins): exceeds > active > insufficient
_RANK = np.zeros(256, dtype=np.uint8)
_RANK[S.NOT_AFFECTED], _RANK[S.NOT_THROTTLED], _RANK[S.INSUFFICIENT], _RANK[S.ACTIVE], _RANK[S.EXCEEDS] = 0, 1, 2, 3, 4
_RANK[255] = 5  # pod-level error
_CODE = np.array([S.NOT_AFFECTED, S.NOT_THROTTLED, S.INSUFFICIENT, S.ACTIVE, S.EXCEEDS, 255], dtype=np.uint8)


def combine_status(matrices) -> np.ndarray:
    """Status matrices [n][T] of the pages -> the cluster's."""
    rank = _RANK[np.asarray(matrices[0])]
    for m in matrices[1:]:
        rank = np.maximum(rank, _RANK[np.asarray(m)])
    return _CODE[rank]


def verdicts(status: np.ndarray) -> np.ndarray:
    """PreFilter verdict per pod from its status row (plugin.go:177-214): error, block, or allow."""
    err = (status == 255).any(axis=1)
    blocked = ((status == S.EXCEEDS) | (status == S.ACTIVE) | (status == S.INSUFFICIENT)).any(axis=1)
    return np.where(err, S.VERDICT_ERROR, np.where(blocked, S.VERDICT_BLOCK, S.VERDICT_ALLOW)).astype(np.uint8)


def combine_reconcile(pages, results) -> list:
    """Per throttle row one dict: used / calc (resourceCounts?, resourceRequests by name), throttled (pod flag + by name),
    calc_updated, error — from the pages' reconcile results (rows aligned: every page holds every throttle)."""
    T = len(pages[0].thr_names)
    out = []
    for i in range(T):
        used, calc, thr_by_name = {}, {}, {}
        for b, r in zip(pages, results):
            for dst, tab in ((used, r.used), (calc, r.calc)):
                d = b.amount_to_dict(tab, i)
                if "resourceCounts" in d:
                    dst["resourceCounts"] = d["resourceCounts"]
                if "resourceRequests" in d:
                    dst.setdefault("resourceRequests", {}).update(d["resourceRequests"])
            for name, dim in b.dims.items():
                if int(r.thrl_has[i]) >> dim & 1:
                    thr_by_name[name] = bool(int(r.thrl_flag[i]) >> dim & 1)
        out.append({"used": used, "calc": calc, "throttled": (bool(results[0].thrl_pod[i]), thr_by_name),
                    "calc_updated": any(bool(r.calc_updated[i]) for r in results),
                    "error": any(bool(r.error[i]) for r in results)})
    return out


def status_manifest(pages, results, i, now_text: str, previous: dict | None = None) -> dict:
    """``status`` of throttle row ``i`` after a paged reconcile, as UpdateStatus would write it (see
    BuiltState.status_manifest): the resource names of all pages in one document."""
    docs = [b.status_manifest(r, i, now_text, previous=None) for b, r in zip(pages, results)]
    st = dict(previous or {})
    used = {}
    for d in docs:
        if "resourceCounts" in d["used"]:
            used["resourceCounts"] = d["used"]["resourceCounts"]
        if "resourceRequests" in d["used"]:
            used.setdefault("resourceRequests", {}).update(d["used"]["resourceRequests"])
    st["used"] = used
    if any(bool(r.calc_updated[i]) for r in results):  # replaced as a whole: every page contributes its names
        thr = {}
        for b, r in zip(pages, results):
            a = b.amount_to_manifest(r.calc, i)
            if "resourceCounts" in a:
                thr["resourceCounts"] = a["resourceCounts"]
            if "resourceRequests" in a:
                thr.setdefault("resourceRequests", {}).update(a["resourceRequests"])
        st["calculatedThreshold"] = {"threshold": thr, "calculatedAt": now_text, "messages": list(pages[0].thr_messages[i])}
    else:
        st.setdefault("calculatedThreshold", {})
    throttled = {"resourceCounts": docs[0]["throttled"]["resourceCounts"], "resourceRequests": {}}
    for d in docs:
        throttled["resourceRequests"].update(d["throttled"]["resourceRequests"])
    st["throttled"] = throttled
    return st


class PagedEngine:
    """One HIP engine per page of a ``ClusterState.build_pages()`` result; reconcile and check run on every page (the
    selector scan is repeated per page: the price of more than 16 resource names) and come back combined."""

    def __init__(self, pages, kernel_variant=E.VARIANT_INDEXED, device=-1):
        self.pages = pages
        self.engines = [E.Engine.for_snapshot(b.snapshot, kernel_variant, device) for b in pages]

    def close(self):
        for e in self.engines:
            e.close()

    def reconcile(self, now, apply=True):
        """-> (combined rows, per-page ReconcileResult list).  The step runs through the library's kt_paged_reconcile (what
        a Go host calls); the rows are put together by resource NAME here, which only the host layer knows."""
        results, replaced_any, error_any = E.paged_reconcile(self.engines, now, apply=apply)
        rows = combine_reconcile(self.pages, results)
        for i, r in enumerate(rows):  # the library's OR over the pages is the same statement
            assert r["calc_updated"] == bool(replaced_any[i]) and r["error"] == bool(error_any[i])
        return rows, results

    def check(self, on_equal=False):
        """-> (status matrix [pods][throttles], verdict per pod) of the whole cluster — combined inside the library
        (kt_paged_check: the C-ABI entry point of this module's combination rule)."""
        n = self.pages[0].snapshot.n_pods
        status, summary = E.paged_check(self.engines, n, on_equal=on_equal)
        v = verdicts(status)
        got = np.where(summary == 2, S.VERDICT_ERROR, np.where((summary & 1) != 0, S.VERDICT_BLOCK, S.VERDICT_ALLOW)).astype(np.uint8)
        assert (got == v).all(), "kt_paged_check: summary words disagree with the combined status rows"
        return status, v
