"""More resource names than an engine has dimensions: ClusterState.build_pages() + kube_throttler_amd/paging.py.

Random clusters whose requests and thresholds draw from 40 resource names (three pages of <= 16) go through the pages —
here with the CPU oracle standing in for the per-page engine, tests/test_paging_gpu.py runs the engines — and the
COMBINED answers must equal the manifest-level model of tests/manifest_model.py, which has no notion of dimensions:
`used`, calculated thresholds (and whether they were replaced), throttled flags, and — after the status was written back —
every (pod, throttle) CheckThrottleStatus and PreFilter verdict for both isThrottledOnEqual values.  The reference sums
any resource name (pkg/resourcelist/resourcelist.go:27-54)."""
import numpy as np
import pytest

import test_manifest_model as TM
from kube_throttler_amd import paging
from kube_throttler_amd import snapshot as S
from kube_throttler_amd.quantity import parse_rfc3339
from manifest_model import Model

MANY = {f"example.com/r{k:02d}": ["0", "1", "2", "3"] for k in range(37)}
MANY_THR = {name: ["1", "2", "4", "6"] for name in MANY}


def wide_cluster(seed):
    """tests/test_manifest_model.random_cluster with 40 resource names to draw from."""
    qty, thr = dict(TM.QTY, **MANY), dict(TM.THR_QTY, **MANY_THR)
    old = TM.QTY, TM.THR_QTY
    TM.QTY, TM.THR_QTY = qty, thr
    try:
        return TM.random_cluster(seed)
    finally:
        TM.QTY, TM.THR_QTY = old


def responsible_rows(snap):
    need = S.THR_VALID | S.THR_RESPONSIBLE
    return np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]


def check_against_model(cs, reconcile_pages, check_pages, label):
    """reconcile_pages(pages, now) -> per-page full-row results; check_pages(pages, on_equal) -> per-page status matrices."""
    model = Model(cs)
    now = parse_rfc3339(TM.NOW_TEXT)
    pages = cs.build_pages()
    assert len(pages) >= 3 and all(b.snapshot.D <= S.KT_MAX_DIMS for b in pages), f"{label}: {len(pages)} pages"
    rows = responsible_rows(pages[0].snapshot)
    results = reconcile_pages(pages, now)
    combined = paging.combine_reconcile(pages, results)
    for i in rows:
        thr = cs.throttles[i]
        want = model.reconcile(thr, now)
        where = f"{label}/{pages[0].thr_names[i]}"
        assert combined[i]["error"] == (want is None), f"{where}: reconcile error"
        if want is None:
            continue
        assert combined[i]["used"] == want["used"].as_dict(), f"{where}: used"
        assert combined[i]["calc"] == want["calc"].as_dict(), f"{where}: calculated threshold"
        assert combined[i]["calc_updated"] == want["updated"], f"{where}: calculatedThreshold replaced"
        assert combined[i]["throttled"] == want["throttled"], f"{where}: throttled"
    for i in rows:
        if not combined[i]["error"]:
            cs.throttles[i]["status"] = paging.status_manifest(pages, results, i, TM.NOW_TEXT, previous=cs.throttles[i].get("status"))
    pages = cs.build_pages()
    name_of = {S.NOT_THROTTLED: "not-throttled", S.ACTIVE: "active", S.INSUFFICIENT: "insufficient",
               S.EXCEEDS: "pod-requests-exceeds-threshold"}
    for on_equal in (False, True):
        status = paging.combine_status(check_pages(pages, on_equal))
        verdict = paging.verdicts(status)
        for i, p in enumerate(cs.pods):
            want_v, want_st = model.check(p, on_equal)
            where = f"{label}/pod{i} on_equal={on_equal}"
            assert {S.VERDICT_ALLOW: "allow", S.VERDICT_BLOCK: "block", S.VERDICT_ERROR: "error"}[int(verdict[i])] == want_v, where
            if want_v == "error":
                continue
            got = {pages[0].thr_names[t]: name_of[int(status[i, t])] for t in range(len(pages[0].thr_names))
                   if status[i, t] != S.NOT_AFFECTED}
            assert got == want_st, where


def full_rows(res, rows, snap):
    """An oracle reconcile of `rows` as a result indexed by throttle row (what an engine returns)."""
    full = type("R", (), {})()
    for f in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
        a = np.zeros(snap.n_thr + 1, getattr(res, f).dtype)
        a[rows] = getattr(res, f)[:len(rows)]
        setattr(full, f, a)
    for tab in ("used", "calc"):
        t = S.Amounts(snap.n_thr + 1, snap.D)
        for f in ("v", "present", "count", "has_count"):
            getattr(t, f)[rows] = getattr(getattr(res, tab), f)[:len(rows)]
        setattr(full, tab, t)
    return full


@pytest.mark.parametrize("seed", range(40))
def test_pages_combined_equal_the_manifest_model(seed, oracle_mod):
    def reconcile_pages(pages, now):
        out = []
        for b in pages:
            rows = responsible_rows(b.snapshot)
            out.append(full_rows(oracle_mod.Oracle(b.snapshot).reconcile(now, rows=rows), rows, b.snapshot))
        return out

    def check_pages(pages, on_equal):
        return [oracle_mod.Oracle(b.snapshot).check(on_equal=on_equal)[0] for b in pages]

    check_against_model(wide_cluster(seed), reconcile_pages, check_pages, f"seed {seed}")


def test_a_cluster_with_few_names_is_one_page():
    cs = TM.random_cluster(3)
    pages = cs.build_pages()
    assert len(pages) == 1 and pages[0].only is None
