"""tests/test_paging_cpu.py with the HIP engines as the per-page evaluators (kube_throttler_amd.paging.PagedEngine)."""
import os

import pytest

from kube_throttler_amd import paging
from test_paging_cpu import check_against_model, wide_cluster

pytestmark = pytest.mark.gpu


# (KT_FUZZ_PAGED=n adds the seeds 100 .. 100 + n: profiles/r06_fuzz.txt)
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 8] + list(range(100, 100 + int(os.environ.get("KT_FUZZ_PAGED", "0")))))
def test_paged_engines_equal_the_manifest_model(seed):
    state = {}

    def reconcile_pages(pages, now):
        eng = paging.PagedEngine(pages)
        try:
            return eng.reconcile(now, apply=False)[1]  # (kt_paged_reconcile; PagedEngine checks the library's OR-ed flags)
        finally:
            eng.close()

    def check_pages(pages, on_equal):
        eng = state.get("eng")
        if eng is None or eng.pages is not pages:
            if eng is not None:
                eng.close()
            eng = state["eng"] = paging.PagedEngine(pages)
        # ONE matrix: the pages combined inside the library (kt_paged_check, the C-ABI form of paging.combine_status) —
        # check_against_model combines what it is handed, and the combination of a single matrix is that matrix
        status, _ = eng.check(on_equal=on_equal)
        # ... which must also be what the Python statement of the rule makes of the per-page matrices
        n = pages[0].snapshot.n_pods
        per_page = [e.check(n=n, on_equal=on_equal, want_status=True)[0] for e in eng.engines]
        assert (paging.combine_status(per_page) == status).all()
        return [status]

    try:
        check_against_model(wide_cluster(seed), reconcile_pages, check_pages, f"seed {seed}")
    finally:
        if state.get("eng") is not None:
            state["eng"].close()
