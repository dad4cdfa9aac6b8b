"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle and the golden vectors.

Bit-exact bar: every per-(pod, throttle) status, every per-pod summary word, every `used` vector,
presence mask, count, calculated threshold and throttled flag must be identical.
"""
import os

import numpy as np
import pytest

from kube_throttler_amd import engine as E
from kube_throttler_amd import snapshot as S
from kube_throttler_amd import workload as W
from scenario_runner import load_scenarios, run_scenario

pytestmark = pytest.mark.gpu

VARIANTS = [E.VARIANT_INDEXED, E.VARIANT_DENSE]
VIDS = ["indexed", "dense"]
NOW = (1767225600, 0)


class EngineBackend:
    def __init__(self, variant):
        self.variant = variant

    def reconcile(self, built, now):
        e = E.Engine.for_snapshot(built.snapshot, self.variant)
        try:
            return e.reconcile(now, apply=False)
        finally:
            e.close()

    def check(self, built, rows, on_equal):
        e = E.Engine.for_snapshot(built.snapshot, self.variant)
        try:
            return e.check(rows, on_equal=on_equal, want_status=True)
        finally:
            e.close()


SCENARIOS = load_scenarios()


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_golden_scenarios(sc, variant):
    """The reference's integration scenarios (G1-G3, G5), on the GPU."""
    run_scenario(sc, EngineBackend(variant))


def assert_reconcile_equal(got, want, T, skip_error_rows=True):
    err_g, err_w = got.error[:T], want.error[:T]
    np.testing.assert_array_equal(err_g != 0, err_w != 0, err_msg="reconcile error flags")
    ok = err_w[:T] == 0
    for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod"):
        np.testing.assert_array_equal(getattr(got, name)[:T][ok], getattr(want, name)[:T][ok], err_msg=name)
    for tab in ("used", "calc"):
        g, w = getattr(got, tab), getattr(want, tab)
        np.testing.assert_array_equal(g.present[:T][ok], w.present[:T][ok], err_msg=f"{tab}.present")
        np.testing.assert_array_equal(g.has_count[:T][ok], w.has_count[:T][ok], err_msg=f"{tab}.has_count")
        np.testing.assert_array_equal(g.count[:T][ok], w.count[:T][ok], err_msg=f"{tab}.count")
        np.testing.assert_array_equal(g.v[:T][ok], w.v[:T][ok], err_msg=f"{tab}.v")


def responsible_rows(snap):
    need = S.THR_VALID | S.THR_RESPONSIBLE
    return np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]


def _rows_of(rec_all, rows, D):
    """the listed throttle rows of a whole-engine reconcile result, in the oracle's layout"""
    got = E.ReconcileResult(len(rows), D)
    for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
        getattr(got, name)[:len(rows)] = getattr(rec_all, name)[rows]
    for tab in ("used", "calc"):
        for f in ("v", "present", "count", "has_count"):
            getattr(getattr(got, tab), f)[:len(rows)] = getattr(getattr(rec_all, tab), f)[rows]
    return got


def run_full_parity(snap, oracle_mod, variant, now=NOW, on_equals=(False, True), nthreads=8):
    """reconcile -> store status -> check, oracle vs engine, everything compared."""
    T = snap.n_thr
    o = oracle_mod.Oracle(snap)
    eng = E.Engine.for_snapshot(snap, variant)
    try:
        # resourcelist.PodRequestResourceList parity
        ov, op = o.pod_requests()
        gv, gp = eng.fetch_pod_requests(n=snap.n_pods)
        np.testing.assert_array_equal(gp, op, err_msg="pod request presence")
        np.testing.assert_array_equal(gv, ov, err_msg="pod request values")
        # reconcile (only responsible throttles are ever reconciled by the reference)
        rows = responsible_rows(snap)
        want = o.reconcile(now, rows=rows, nthreads=nthreads)
        got_all = eng.reconcile(now, apply=True)
        got = _rows_of(got_all, rows, snap.D)
        assert_reconcile_equal(got, want, len(rows))
        # NextOverrideHappensIn (the controller's enqueueAfter instant) of every reconciled throttle
        ws, wn, wh = o.next_override(now)
        gs, gn, gh = eng.next_override()
        okr = rows[want.error[:len(rows)] == 0]
        np.testing.assert_array_equal(gh[okr], wh[okr], err_msg="next override: has")
        np.testing.assert_array_equal(gs[okr], ws[okr], err_msg="next override: seconds")
        np.testing.assert_array_equal(gn[okr], wn[okr], err_msg="next override: nanoseconds")
        # UpdateStatus on the oracle side, then check against the stored status
        snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod,
                          want.error, rows=rows)
        for on_equal in on_equals:
            st_w, sm_w = o.check(on_equal=on_equal, nthreads=nthreads)
            st_g, sm_g = eng.check(n=snap.n_pods, on_equal=on_equal, want_status=True)
            np.testing.assert_array_equal(st_g, st_w, err_msg=f"status matrix on_equal={on_equal}")
            np.testing.assert_array_equal(sm_g, sm_w, err_msg=f"summary on_equal={on_equal}")
            # the PreFilter sweep proper (summary words only: the lean / wordwise instantiations of the check kernel)
            _, sm_l = eng.check(n=snap.n_pods, on_equal=on_equal, want_status=False)
            np.testing.assert_array_equal(sm_l, sm_w, err_msg=f"summary of the lean sweep on_equal={on_equal}")
        ret = (st_w, sm_w, want)
        # kt_sweep_launch: that sweep (against the status stored so far) AND a reconcile as one pass over the pod tables —
        # the fused kernel where the program allows it, the two launches one after the other otherwise; either way the
        # results of the pair
        for on_equal in on_equals:
            _, sm_w2 = o.check(on_equal=on_equal, want_status=False, nthreads=nthreads)
            want2 = o.reconcile(now, rows=rows, nthreads=nthreads)
            eng.sweep_launch(now, apply=True, on_equal=on_equal)
            _, sm_s = eng.check_fetch(snap.n_pods, False)
            np.testing.assert_array_equal(sm_s, sm_w2, err_msg=f"summary of kt_sweep_launch on_equal={on_equal}")
            assert_reconcile_equal(_rows_of(eng.reconcile_fetch(), rows, snap.D), want2, len(rows))
            snap.apply_status(want2.used, want2.calc, want2.calc_updated, want2.thrl_flag, want2.thrl_has, want2.thrl_pod,
                              want2.error, rows=rows)
        return ret
    finally:
        eng.close()


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_small_rich(seed, variant, oracle_mod):
    """Every selector operator, multi-term OR, overrides, reserved amounts, both kinds."""
    snap = W.generate(W.small(seed=seed))
    st, sm, rec = run_full_parity(snap, oracle_mod, variant)
    assert set(np.unique(st)) >= {S.NOT_AFFECTED, S.NOT_THROTTLED, S.ACTIVE}


@pytest.mark.parametrize("shape", ["packed", "plain-negative", "sixteen-dims", "multiterm-rich", "incremental", "chunked"])
def test_aggregate_rank_windows(shape, oracle_mod, monkeypatch):
    """The aggregate's table of per-throttle records does not have to fit LDS at once (round 6): the kernel scans a chunk once per
    WINDOW of ranks and folds only the matches whose record the table holds at that time — what keeps a 16-dimension engine's
    program in ONE index chunk (tests/test_engine_gpu.py::test_sixteen_dims_sixteen_labels_1m runs it at full size).  Here
    KT_AGG_SMALL_WINDOW forces windows of 64 records on small clusters (150-400 ranks: three to seven passes) in every fold form —
    the packed word queue, the plain fold (a negative request), 16 dimensions, throttles with several terms (the run rule across a
    window boundary), an incremental engine's delta scans, a multi-chunk index — and everything is compared with the oracle."""
    monkeypatch.setenv("KT_AGG_SMALL_WINDOW", "1")
    if shape == "chunked":
        monkeypatch.setenv("KT_CHUNK_BUDGET", "9000")
    cfg = {"packed": W.small(seed=61, n_pods=5000, n_thr=300, n_cluster=150),
           "plain-negative": W.small(seed=62, n_pods=5000, n_thr=240, n_cluster=120),
           "sixteen-dims": W.small(seed=63, n_pods=4000, n_thr=200, n_cluster=100, D=16),
           "multiterm-rich": W.small(seed=64, n_pods=4000, n_thr=160, n_cluster=80, terms=(1, 4), reqs=(1, 3)),
           "incremental": W.small(seed=65, n_pods=3000, n_thr=200, n_cluster=100),
           "chunked": W.small(seed=66, n_pods=4000, n_thr=260, n_cluster=130, terms=(1, 3), reqs=(1, 3))}[shape]
    snap = W.generate(cfg)
    if shape == "plain-negative":  # one negative request: the engine leaves the packed fold for good
        snap.ctr_req[int(snap.pod_ctr_off[7]), 0] = -5
    variant = E.VARIANT_INDEXED | (E.VARIANT_INCREMENTAL if shape == "incremental" else 0)
    st, sm, rec = run_full_parity(snap, oracle_mod, variant)
    assert (st != S.NOT_AFFECTED).any()
    if shape == "incremental":  # the maintained partials: pods rewritten in place go out and in through windowed delta scans
        eng = E.Engine.for_snapshot(snap, variant)
        try:
            eng.reconcile(NOW, apply=False)
            rows = np.arange(0, snap.n_pods, 7, dtype=np.int64)
            eng.upsert_pods(_permute_pods(snap, rows), rows=rows)
            o = oracle_mod.Oracle(snap)
            rows_t = responsible_rows(snap)
            want = o.reconcile(NOW, rows=rows_t)
            got = eng.reconcile(NOW, apply=False)
            np.testing.assert_array_equal(got.used.v[rows_t], want.used.v[:len(rows_t)])
            np.testing.assert_array_equal(got.used.count[rows_t], want.used.count[:len(rows_t)])
        finally:
            eng.close()


@pytest.mark.parametrize("switch", ["KT_NO_NS_ORDER", "KT_NO_SCAN_VIEW"])
def test_row_list_scans_of_a_multi_chunk_program(switch, oracle_mod, monkeypatch):
    """A program of several chunks is cut for the packed fold's records — but a full scan that gathers through the row list instead of
    streaming the scan view (the A/B switches KT_NO_NS_ORDER / KT_NO_SCAN_VIEW) folds PLAIN records: the engine must plan no packed
    fold then and cut the chunks for plain records.  Found by tests/test_fuzz_gpu.py in round 6 (the reconcile answered "a chunk of
    the selector index exceeds the aggregate kernel's LDS budget": the plan was made for a scan that never used it)."""
    monkeypatch.setenv("KT_CHUNK_BUDGET", "9000")
    monkeypatch.setenv(switch, "1")
    snap = W.generate(W.small(seed=97, n_pods=4000, n_thr=160, n_cluster=40, D=5, n_ns=11, K=12, V=4, L=8, terms=(1, 3), reqs=(0, 4)))
    run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
    try:
        eng.reconcile(NOW, apply=False)
        assert eng.index_stats()["chunks"] > 1 and eng.packed_words() == 0
        assert not eng.kernel_name(E.KERNEL_AGGREGATE).startswith("kt_aggregate_bitmap_packed")
    finally:
        eng.close()


@pytest.mark.parametrize("shape", ["simple", "rich-8-labels", "rich-16-labels", "rich-32-labels", "five-keys", "windows", "chunked", "nine-dims"])
def test_packed_fold_of_more_than_four_words(shape, oracle_mod, monkeypatch):
    """An engine with more than 8 dimensions packs its requests into up to EIGHT 64-bit words (round 6: until then `PackPlan` held
    four and a 16-dimension engine ran the plain fold — 144-byte records, up to 17 LDS atomics per match).  The NW = 8
    instantiations of kt_aggregate_bitmap in every form that exists — no veto columns, 8 / 16 / 32 atom slots, terms with four
    or five positive keys, rank windows, a multi-chunk index — and the reduction / fused finalize over 72-byte records: the
    engine must report the packed kernel with 5..8 words, and everything is compared with the oracle."""
    if shape == "windows":
        monkeypatch.setenv("KT_AGG_SMALL_WINDOW", "1")
    if shape == "chunked":
        monkeypatch.setenv("KT_CHUNK_BUDGET", "12000")
    cfg = {"simple": W.small(seed=801, n_pods=6000, n_thr=96, n_cluster=48, D=16, rich_ops=0, terms=(1, 1), reqs=(1, 2)),
           "rich-8-labels": W.small(seed=802, n_pods=6000, n_thr=120, n_cluster=60, D=16, L=6, K=12),
           "rich-16-labels": W.small(seed=803, n_pods=6000, n_thr=120, n_cluster=60, D=16, L=14, K=24),
           "rich-32-labels": W.small(seed=804, n_pods=5000, n_thr=100, n_cluster=50, D=16, L=28, K=40),
           "five-keys": W.small(seed=805, n_pods=8000, n_thr=100, n_cluster=50, D=16, n_ns=8, K=16, V=3, L=10, terms=(1, 3), reqs=(3, 5)),
           "windows": W.small(seed=806, n_pods=5000, n_thr=300, n_cluster=150, D=16, terms=(1, 4), reqs=(1, 3)),
           "chunked": W.small(seed=807, n_pods=5000, n_thr=260, n_cluster=130, D=16, terms=(1, 3), reqs=(1, 3)),
           "nine-dims": W.small(seed=808, n_pods=5000, n_thr=90, n_cluster=45, D=9)}[shape]
    snap = W.generate(cfg)
    if shape == "nine-dims":  # nine wide fields: more than four words although the engine has barely more than 8 dimensions
        n_ctr = int(snap.pod_ctr_off[snap.n_pods])
        snap.ctr_present[:n_ctr] |= 0x1FF  # (every container names all nine resources)
        base = snap.ctr_req[:n_ctr, :9] & 0xFFF
        for shift in range(3, 16):  # the narrowest fields that no longer fit four words
            snap.ctr_req[:n_ctr, :9] = (base << shift) | 1
            probe = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
            try:
                probe.reconcile(NOW, apply=False)
                if probe.packed_words() > 4:
                    break
            finally:
                probe.close()
    st, sm, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    assert (st != S.NOT_AFFECTED).any() and (rec.used.v != 0).any()
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
    try:
        rows = responsible_rows(snap)
        want = oracle_mod.Oracle(snap).reconcile(NOW, rows=rows)
        got = eng.reconcile(NOW, apply=False)
        assert eng.kernel_name(E.KERNEL_AGGREGATE).startswith("kt_aggregate_bitmap_packed"), eng.kernel_name(E.KERNEL_AGGREGATE)
        assert 4 < eng.packed_words() <= 8, eng.packed_words()
        if shape == "chunked":
            assert eng.index_stats()["chunks"] > 1
        assert_reconcile_equal(_rows_of(got, rows, snap.D), want, len(rows))
        # the separate reduction over the same slabs (kt_aggregate_launch -> kt_finalize_launch: kt_reduce_packed_slabs)
        eng.aggregate_launch()
        eng.finalize_launch(NOW, apply=False)
        assert_reconcile_equal(_rows_of(eng.reconcile_fetch(), rows, snap.D), want, len(rows))
    finally:
        eng.close()


def test_chunks_are_cut_again_when_the_plain_fold_is_needed(oracle_mod, monkeypatch):
    """A program of several index chunks is cut for the packed fold's 40-byte records (round 6: more words per chunk).  The first
    scan that needs the PLAIN fold — here a pod with a negative request arrives after the program was compiled — finds tables and
    slab areas too small for its 80-byte records: the engine cuts the index again for plain records (one recompile, the launch that
    noticed upgrades its lock) and every result stays the oracle's, before and after."""
    monkeypatch.setenv("KT_CHUNK_BUDGET", "9000")
    snap = W.generate(W.small(seed=67, n_pods=4000, n_thr=260, n_cluster=130, terms=(1, 3), reqs=(1, 3)))
    o = oracle_mod.Oracle(snap)
    rows_t = responsible_rows(snap)
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
    try:
        got = eng.reconcile(NOW, apply=False)
        want = o.reconcile(NOW, rows=rows_t)
        np.testing.assert_array_equal(got.used.v[rows_t], want.used.v[:len(rows_t)])
        assert eng.index_stats()["chunks"] > 1 and eng.kernel_name(E.KERNEL_AGGREGATE).startswith("kt_aggregate_bitmap_packed")
        compiles = eng.compiles()
        # a counted pod gets a negative request (resource.Quantity allows it; the packed words do not)
        counted = np.nonzero((snap.pod_flags & (S.POD_VALID | S.POD_SCHED_MATCH | S.POD_SCHEDULED | S.POD_FINISHED)) ==
                             (S.POD_VALID | S.POD_SCHED_MATCH | S.POD_SCHEDULED))[0]
        row = int(counted[3])
        snap.ctr_req[int(snap.pod_ctr_off[row]), 0] = -7
        eng.upsert_pods(_permute_pods(snap, np.array([row])), rows=np.array([row], dtype=np.int64))
        o2 = oracle_mod.Oracle(snap)
        got2 = eng.reconcile(NOW, apply=True)
        want2 = o2.reconcile(NOW, rows=rows_t)
        np.testing.assert_array_equal(got2.used.v[rows_t], want2.used.v[:len(rows_t)])
        np.testing.assert_array_equal(got2.used.count[rows_t], want2.used.count[:len(rows_t)])
        np.testing.assert_array_equal(got2.thrl_flag[rows_t], want2.thrl_flag[:len(rows_t)])
        assert eng.compiles() == compiles + 1 and not eng.kernel_name(E.KERNEL_AGGREGATE).startswith("kt_aggregate_bitmap_packed")
        snap.apply_status(want2.used, want2.calc, want2.calc_updated, want2.thrl_flag, want2.thrl_has, want2.thrl_pod, want2.error, rows=rows_t)
        _, sm_w = o2.check(want_status=False)
        _, sm_g = eng.check(n=snap.n_pods, want_status=False)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()


@pytest.mark.parametrize("shape", ["simple-multiterm", "rich-16-dims", "simple-12-dims", "rich-one-per-cu"])
def test_lean_sweep_shapes(shape, oracle_mod, monkeypatch):
    """The PreFilter sweep (summary words only) settles matches per 64-bit word with the WordVerdict masks and the
    per-nibble `active` tables: the simple and the rich instantiation, two workgroups per CU and one, 8 and 16 dimension
    slots (two / four nibbles), throttles with several terms (run masks) — run_full_parity compares the lean sweep too."""
    cfg = {"simple-multiterm": W.small(seed=31, n_pods=3000, n_thr=96, n_cluster=48, rich_ops=0, terms=(1, 3), reqs=(1, 2)),
           "rich-16-dims": W.small(seed=32, n_pods=3000, n_thr=96, n_cluster=48, D=16),
           "simple-12-dims": W.small(seed=33, n_pods=3000, n_thr=96, n_cluster=48, D=12, rich_ops=0, terms=(1, 2), reqs=(1, 2)),
           "rich-one-per-cu": W.small(seed=34, n_pods=3000, n_thr=96, n_cluster=48)}[shape]
    if shape == "rich-one-per-cu":
        monkeypatch.setenv("KT_CHECK_ONE_PER_CU", "1")
    run_full_parity(W.generate(cfg), oracle_mod, E.VARIANT_INDEXED)


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_random_selector_errors(variant, oracle_mod):
    """Unconvertible podSelector terms (pod-level Error, reconcile error), swallowed namespaceSelector
    errors, pods in namespaces without a Namespace object."""
    # one bad namespaced Throttle: only the pods of its namespace that reach the bad term are in Error
    snap = W.generate(W.small(seed=7, n_pods=3000, n_thr=96, n_cluster=48, n_invalid_pod_sel=1, n_invalid_ns_sel=4,
                              n_missing_ns=2))
    st, sm, rec = run_full_parity(snap, oracle_mod, variant)
    verdict = S.summary_fields(sm)[0]
    assert (verdict == S.VERDICT_ERROR).any() and (verdict != S.VERDICT_ERROR).any()
    assert rec.error.any()
    # bad terms on ClusterThrottles too (their empty namespaceSelector reaches every namespace)
    snap = W.generate(W.small(seed=8, n_pods=2000, n_thr=96, n_cluster=48, n_invalid_pod_sel=6, n_invalid_ns_sel=2))
    st, sm, rec = run_full_parity(snap, oracle_mod, variant)
    assert (S.summary_fields(sm)[0] == S.VERDICT_ERROR).any() and rec.error.any()


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_edge_shapes(variant, oracle_mod):
    """D=1, D=16, L=16, terms without requirements, throttles without terms, tiny pod counts."""
    for cfg in (W.small(seed=11, n_pods=1, n_thr=3, n_cluster=1, D=1, n_ns=1, K=2, V=2, L=1, terms=(0, 1), reqs=(0, 1)),
                W.small(seed=12, n_pods=65, n_thr=40, n_cluster=20, D=16, n_ns=3, K=16, V=2, L=16, terms=(0, 3), reqs=(0, 4)),
                W.small(seed=13, n_pods=300, n_thr=33, n_cluster=0, D=3, n_ns=2, K=4, V=2, L=2, terms=(1, 2), reqs=(0, 2),
                        overrides=0)):
        snap = W.generate(cfg)
        run_full_parity(snap, oracle_mod, variant)


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_wide_label_sets(variant, oracle_mod):
    """Pods with more labels than the scan kernels keep atom slots for: the reference matches labels.Set(pod.Labels) of
    any size (throttle_selector.go:48-54).  24 labels over 24 keys (every label relevant: 16-/32-slot atom rows), 40 labels
    over 48 keys with selectors touching most keys (pods whose relevant atoms overflow 32 slots walk the raw labels),
    and 40 labels of which the selectors reference only a few keys (the irrelevant ones are dropped at translation)."""
    for cfg in (W.small(seed=21, n_pods=1500, n_thr=64, n_cluster=32, D=4, n_ns=4, K=24, V=2, L=24, terms=(1, 3), reqs=(1, 3)),
                W.small(seed=22, n_pods=1500, n_thr=160, n_cluster=96, D=4, n_ns=4, K=48, V=2, L=40, terms=(1, 3), reqs=(1, 4)),
                W.small(seed=23, n_pods=1500, n_thr=6, n_cluster=2, D=4, n_ns=4, K=64, V=2, L=40, terms=(1, 2), reqs=(1, 2))):
        snap = W.generate(cfg)
        assert snap.L > 16
        st, sm, rec = run_full_parity(snap, oracle_mod, variant)
        assert (st == S.NOT_THROTTLED).any() or (st == S.ACTIVE).any()


@pytest.mark.parametrize("variant", VARIANTS, ids=VIDS)
def test_config1_full(variant, oracle_mod):
    """BASELINE.json configs[1]: 10k pods x 100 Throttles, D=4, single selectorTerm — full matrix."""
    snap = W.generate(W.preset(1))
    run_full_parity(snap, oracle_mod, variant)


def test_empty_engine(oracle_mod):
    """No throttles / no pods: everything is allowed, nothing crashes."""
    snap = W.generate(W.small(seed=5, n_pods=10, n_thr=0, n_cluster=0))
    eng = E.Engine.for_snapshot(snap)
    st, sm = eng.check(n=10, want_status=False)
    assert (sm == 0).all()
    r = eng.reconcile(NOW)
    eng.close()


def test_incremental_feed_matches_bulk(oracle_mod):
    """Upserting pods/throttles/namespaces row by row (out of order, with overwrites and deletes) ends
    in the same state as a bulk load of the final snapshot."""
    final = W.generate(W.small(seed=21, n_pods=1500, n_thr=48, n_cluster=24))
    other = W.generate(W.small(seed=22, n_pods=1500, n_thr=48, n_cluster=24))
    eng = E.Engine(final.D, final.L, 2000, 64, 16)
    try:
        eng.upsert_namespaces(other)
        eng.upsert_throttles(other)
        eng.upsert_pods(other)                      # garbage first ...
        eng.reconcile(NOW, apply=True)              # ... including a device-resident status
        eng.upsert_namespaces(final)                # then the real content, pods in reverse row order
        rows = np.arange(final.n_pods)[::-1].copy()
        rev = _permute_pods(final, rows)
        eng.upsert_pods(rev, rows=rows)
        eng.upsert_throttles(final)
        extra = np.array([1700, 1701], dtype=np.int64)
        eng.upsert_pods(_permute_pods(other, np.array([0, 1])), rows=extra)
        eng.delete_pods(extra)
        o = oracle_mod.Oracle(final)
        rows_t = responsible_rows(final)
        want = o.reconcile(NOW, rows=rows_t)
        got = eng.reconcile(NOW, apply=True)
        np.testing.assert_array_equal(got.used.v[rows_t], want.used.v[:len(rows_t)])
        np.testing.assert_array_equal(got.used.count[rows_t], want.used.count[:len(rows_t)])
        final.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod,
                           want.error, rows=rows_t)
        st_w, sm_w = o.check()
        st_g, sm_g = eng.check(n=final.n_pods, want_status=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
        # explicit row lists (the n=1 drop-in shape of PreFilter)
        pick = np.array([5, 999, 17, 5], dtype=np.int64)
        st_g, sm_g = eng.check(rows=pick, want_status=True)
        np.testing.assert_array_equal(st_g, st_w[pick])
        np.testing.assert_array_equal(sm_g, sm_w[pick])
    finally:
        eng.close()


@pytest.mark.parametrize("shape", ["default", "wide", "sixteen-dims"])
def test_event_sized_batches_match_bulk(shape, oracle_mod, monkeypatch):
    """Pods fed as informer events — batches of one to four, in scrambled row order, every row twice (garbage first) — end
    in the state a bulk load gives: kt_feed_few (one wave per pod: lane = dimension / label slot, the batch read from LDS)
    against kt_ingest_pods + kt_translate_pods.  The generator gives init containers, overhead, explicit zeros and absent
    keys; `wide` carries more relevant labels than atom slots (overflow pods), `sixteen-dims` 16 dimension slots."""
    cfg = {"default": W.small(seed=41, n_pods=400, n_thr=48, n_cluster=24),
           "wide": W.small(seed=42, n_pods=300, n_thr=160, n_cluster=96, D=4, n_ns=4, K=48, V=2, L=40, terms=(1, 3), reqs=(1, 4)),
           "sixteen-dims": W.small(seed=43, n_pods=300, n_thr=48, n_cluster=24, D=16)}[shape]
    final = W.generate(cfg)
    cfg2 = W.small(seed=cfg.seed + 100, n_pods=cfg.n_pods_total, n_thr=cfg.n_thr, n_cluster=cfg.n_cluster, D=cfg.D, n_ns=cfg.n_ns, K=cfg.K, V=cfg.V,
                   L=cfg.L)
    other = W.generate(cfg2)
    bulk = E.Engine.for_snapshot(final, E.VARIANT_INDEXED)
    eng = E.Engine(final.D, final.L, final.n_pods + 8, final.n_thr + 4, final.n_ns + 4)
    try:
        eng.upsert_namespaces(final)
        eng.upsert_throttles(final)
        eng.reconcile(NOW, apply=False)   # compiles the program: the events below are translated as they arrive
        rng = np.random.default_rng(7)
        for snap in (other, final):
            order = rng.permutation(final.n_pods)
            k = 0
            while k < len(order):
                nb = int(rng.integers(1, 5))
                rows = np.sort(order[k:k + nb]).astype(np.int64)
                eng.upsert_pods(_permute_pods(snap, rows), rows=rows)
                k += nb
        gv, gp = eng.fetch_pod_requests(n=final.n_pods)
        bv, bp = bulk.fetch_pod_requests(n=final.n_pods)
        np.testing.assert_array_equal(gp, bp, err_msg="pod request presence")
        np.testing.assert_array_equal(gv, bv, err_msg="pod request values")
        o = oracle_mod.Oracle(final)
        rows_t = responsible_rows(final)
        want = o.reconcile(NOW, rows=rows_t)
        got = eng.reconcile(NOW, apply=True)
        np.testing.assert_array_equal(got.used.v[rows_t], want.used.v[:len(rows_t)])
        np.testing.assert_array_equal(got.used.present[rows_t], want.used.present[:len(rows_t)])
        np.testing.assert_array_equal(got.used.count[rows_t], want.used.count[:len(rows_t)])
        final.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows_t)
        st_w, sm_w = o.check()
        st_g, sm_g = eng.check(n=final.n_pods, want_status=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()
        bulk.close()


@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("size", ["event", "coalesced", "bulk"])
def test_rows_named_twice_in_one_batch(size, incremental, oracle_mod):
    """Coalesced informer events: ONE kt_upsert_pods batch names pod rows two or three times with different labels, requests and
    phases (Add, then Update of the same pod — throttle_controller.go:400-536 sees them one by one).  The LAST entry of a row
    wins, in every feed path — kt_feed_few (<= 4 entries), kt_feed_small (<= 256), the staged bulk ingest — and for an incremental
    engine, whose delta scans must take the earlier entry out again (the unfused path let two threads write one row)."""
    final = W.generate(W.small(seed=57, n_pods=600, n_thr=64, n_cluster=32, n_invalid_pod_sel=1))
    other = W.generate(W.small(seed=58, n_pods=600, n_thr=64, n_cluster=32))
    P = final.n_pods
    variant = E.VARIANT_INDEXED | (E.VARIANT_INCREMENTAL if incremental else 0)
    eng = E.Engine(final.D, final.L, P + 8, final.n_thr + 4, final.n_ns + 4, -1, variant)
    try:
        eng.upsert_namespaces(final)
        eng.upsert_throttles(final)
        eng.reconcile(NOW, apply=False)  # compiles the program: the batches below are translated (and delta-scanned) as they arrive
        rng = np.random.default_rng(57)
        per = {"event": 2, "coalesced": 40, "bulk": P}[size]   # distinct rows per batch; every batch names most of them several times
        order = rng.permutation(P)
        for k in range(0, P, per):
            mine = order[k:k + per]
            # entries: garbage (the OTHER cluster's pod at that row) for a random subset, possibly twice, then the final content —
            # scrambled, except that a row's final entry comes after its garbage
            entries = []
            for r in mine:
                for _ in range(int(rng.integers(0, 3))):
                    entries.append((float(rng.random()) * 0.9, int(r), False))
                entries.append((0.9 + float(rng.random()) * 0.1, int(r), True))
            entries.sort()
            rows = np.array([r for _, r, _ in entries], dtype=np.int64)
            eng.upsert_pods(_gather_pods([(final if f else other, r) for _, r, f in entries]), rows=rows)
        bulk = E.Engine.for_snapshot(final, E.VARIANT_INDEXED)
        try:
            gv, gp = eng.fetch_pod_requests(n=P)
            bv, bp = bulk.fetch_pod_requests(n=P)
            np.testing.assert_array_equal(gp, bp, err_msg="pod request presence")
            np.testing.assert_array_equal(gv, bv, err_msg="pod request values")
        finally:
            bulk.close()
        o = oracle_mod.Oracle(final)
        rows_t = responsible_rows(final)
        want = o.reconcile(NOW, rows=rows_t)
        got = eng.reconcile(NOW, apply=True)
        np.testing.assert_array_equal(got.used.v[rows_t], want.used.v[:len(rows_t)])
        np.testing.assert_array_equal(got.used.present[rows_t], want.used.present[:len(rows_t)])
        np.testing.assert_array_equal(got.used.count[rows_t], want.used.count[:len(rows_t)])
        final.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows_t)
        st_w, sm_w = o.check()
        st_g, sm_g = eng.check(n=P, want_status=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()


def _with_pods(base, src_rows):
    """A copy of `base` (same namespaces / throttles) whose pod row r holds base's pod src_rows[r] (-1: row deleted)."""
    import copy
    src_rows = np.asarray(src_rows)
    live = np.where(src_rows >= 0, src_rows, 0)
    pods = _permute_pods(base, live)
    out = copy.copy(base)
    for f in ("n_pods", "pod_ns", "pod_flags", "pod_label_off", "pod_label_key", "pod_label_pair", "pod_ctr_off", "ctr_init",
              "ctr_present", "ctr_req", "pod_ovh_present", "pod_ovh"):
        setattr(out, f, getattr(pods, f))
    out.pod_flags = out.pod_flags.copy()
    out.pod_flags[:len(src_rows)][src_rows < 0] = 0
    return out


@pytest.mark.parametrize("budget", [None, 6000])
def test_incremental_event_path(budget, oracle_mod, monkeypatch):
    """KT_VARIANT_INCREMENTAL: pod adds / updates (labels, requests, phase) / deletes keep the `used`
    partials current by delta scans; every reconcile equals a full rescan of the current pods (= the oracle)."""
    if budget:
        monkeypatch.setenv("KT_CHUNK_BUDGET", str(budget))
    base = W.generate(W.small(seed=71, n_pods=2400, n_thr=96, n_cluster=48, n_invalid_pod_sel=1))
    P = 2000
    rng = np.random.default_rng(71)
    state = np.full(P, -1, dtype=np.int64)   # pod row -> which of base's pods it currently holds
    state[:1200] = np.arange(1200)
    eng = E.Engine(base.D, max(base.L, 1), P, max(base.n_thr, 1), max(base.n_ns, 1), -1, E.VARIANT_INDEXED | E.VARIANT_INCREMENTAL)
    try:
        eng.upsert_namespaces(base)
        eng.upsert_throttles(base)
        eng.upsert_pods(_permute_pods(base, np.arange(1200)), rows=np.arange(1200))

        def check_state(expect_scan):
            snap = _with_pods(base, state)
            want = oracle_mod.Oracle(snap).reconcile(NOW, rows=responsible_rows(snap))
            got_all = eng.reconcile(NOW, apply=False)
            rows = responsible_rows(snap)
            got = E.ReconcileResult(len(rows), snap.D)
            for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
                getattr(got, name)[:len(rows)] = getattr(got_all, name)[rows]
            for tab in ("used", "calc"):
                for f in ("v", "present", "count", "has_count"):
                    getattr(getattr(got, tab), f)[:len(rows)] = getattr(getattr(got_all, tab), f)[rows]
            assert_reconcile_equal(got, want, len(rows))
            assert ("no scan" in eng.kernel_name(E.KERNEL_AGGREGATE)) == (not expect_scan)

        check_state(expect_scan=True)            # first reconcile: full scan, baseline
        check_state(expect_scan=False)           # nothing changed: a copy
        # adds
        add_rows = np.arange(1200, 1700)
        state[add_rows] = np.arange(1200, 1700)
        eng.upsert_pods(_permute_pods(base, state[add_rows]), rows=add_rows)
        check_state(expect_scan=False)
        # updates: the rows take the labels / requests / phase of other pods
        upd_rows = rng.choice(1700, 400, replace=False)
        state[upd_rows] = rng.integers(1700, 2400, 400)
        eng.upsert_pods(_permute_pods(base, state[upd_rows]), rows=upd_rows)
        check_state(expect_scan=False)
        # deletes, then re-adding two of the deleted rows
        del_rows = rng.choice(1700, 300, replace=False).astype(np.int64)
        state[del_rows] = -1
        eng.delete_pods(del_rows)
        check_state(expect_scan=False)
        back = del_rows[:2]
        state[back] = [5, 6]
        eng.upsert_pods(_permute_pods(base, state[back]), rows=back)
        check_state(expect_scan=False)
        # Throttle events that leave the selectors alone (status updates, threshold edits) keep the partials ...
        eng.upsert_throttles(base)
        check_state(expect_scan=False)
        # ... a selector-level change voids them: one rescan, then incremental again
        r0 = int(responsible_rows(base)[0])
        base.thr_flags[r0] &= 0xFFFFFFFF ^ S.THR_RESPONSIBLE
        eng.upsert_throttles(base.throttle_batch([r0]), rows=np.array([r0], dtype=np.int32))
        check_state(expect_scan=True)
        state[7] = 2399
        eng.upsert_pods(_permute_pods(base, state[7:8]), rows=np.array([7]))
        check_state(expect_scan=False)
    finally:
        eng.close()


@pytest.mark.parametrize("budget,dims", [(None, 8), (5000, 8), (None, 16), (9000, 16)])
def test_pod_events_between_sweeps(budget, dims, oracle_mod, monkeypatch):
    """The scan lists and scan-ordered record copies (namespace order with a multi-chunk index) are rebuilt after pod
    events: adds, updates that move pods to other namespaces / labels / requests, deletes and a throttle change between
    full sweeps of a NON-incremental engine — every reconcile + check equals the oracle on the pods currently held.
    (16 dimensions: the view records carry eight packed words — kt_patch_scan_views and the rebuilds with that stride.)"""
    if budget:
        monkeypatch.setenv("KT_CHUNK_BUDGET", str(budget))
    base = W.generate(W.small(seed=73, n_pods=2600, n_thr=96, n_cluster=48, D=dims))
    # pod 2599 carries a cpu request larger than every other and with odd low bits: when it arrives, the packed request
    # words of the scan view (PackPlan: field widths / common trailing zeros proved per view) no longer hold — rebuild
    k99 = int(base.pod_ctr_off[2599])
    base.ctr_req[k99, 0] = 3_000_001
    base.ctr_present[k99] |= 1
    P = 2200
    rng = np.random.default_rng(73)
    state = np.full(P, -1, dtype=np.int64)
    state[:1500] = np.arange(1500)
    eng = E.Engine(base.D, max(base.L, 1), P, max(base.n_thr, 1), max(base.n_ns, 1))
    try:
        eng.upsert_namespaces(base)
        eng.upsert_throttles(base)
        eng.upsert_pods(_permute_pods(base, np.arange(1500)), rows=np.arange(1500))

        def sweep():
            n = int(np.nonzero(state >= 0)[0].max()) + 1   # rows in use
            snap = _with_pods(base, state[:n])
            o = oracle_mod.Oracle(snap)
            rows = responsible_rows(snap)
            want = o.reconcile(NOW, rows=rows)
            got_all = eng.reconcile(NOW, apply=True)
            got = E.ReconcileResult(len(rows), snap.D)
            for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
                getattr(got, name)[:len(rows)] = getattr(got_all, name)[rows]
            for tab in ("used", "calc"):
                for f in ("v", "present", "count", "has_count"):
                    getattr(getattr(got, tab), f)[:len(rows)] = getattr(getattr(got_all, tab), f)[rows]
            assert_reconcile_equal(got, want, len(rows))
            snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows)
            st_w, sm_w = o.check(on_equal=False, nthreads=8)
            st_g, sm_g = eng.check(n=n, on_equal=False, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            _, sm_lean = eng.check(n=n, on_equal=False, want_status=False)   # the sweep instantiation (summary words only)
            np.testing.assert_array_equal(sm_lean, sm_w)
            # the stored status is the engine's: keep base's copy in step for the next round's oracle
            base.thr_used, base.thr_calc = snap.thr_used, snap.thr_calc
            base.thr_flags, base.thr_thrl_flag, base.thr_thrl_has = snap.thr_flags, snap.thr_thrl_flag, snap.thr_thrl_has

        sweep()
        if dims > 8:
            assert eng.packed_words() > 4, eng.packed_words()
        add_rows = np.arange(1500, 2100)
        state[add_rows] = np.arange(1500, 2100)
        eng.upsert_pods(_permute_pods(base, state[add_rows]), rows=add_rows)
        sweep()
        upd_rows = rng.choice(2100, 500, replace=False)
        state[upd_rows] = rng.integers(2100, 2600, 500)      # other namespaces, labels, requests, phases
        eng.upsert_pods(_permute_pods(base, state[upd_rows]), rows=upd_rows)
        sweep()
        del_rows = rng.choice(2000, 400, replace=False).astype(np.int64)
        state[del_rows] = -1
        eng.delete_pods(del_rows)
        sweep()
        # single-pod events: the scan lists / views are patched in place (records rewritten, appended, left behind as
        # "not countable") instead of being rebuilt — kt_patch_scan_views
        for k in range(60):
            r = int(rng.integers(0, 2200))
            if rng.random() < .25 and state[r] >= 0:
                state[r] = -1
                eng.delete_pods(np.array([r], dtype=np.int64))
            else:
                state[r] = int(rng.integers(0, 2599))
                eng.upsert_pods(_permute_pods(base, state[[r]]), rows=np.array([r]))
            if k % 12 == 11:
                sweep()
        state[7] = 2599                                        # does not fit the packed fields: the views are rebuilt
        eng.upsert_pods(_permute_pods(base, state[[7]]), rows=np.array([7]))
        sweep()
        state[8] = 11
        eng.upsert_pods(_permute_pods(base, state[[8]]), rows=np.array([8]))
        sweep()
        eng.upsert_throttles(base)                             # selectors unchanged: the throttle tables go up again, nothing else
        sweep()
        r0 = int(responsible_rows(base)[0])
        base.thr_flags[r0] &= 0xFFFFFFFF ^ S.THR_RESPONSIBLE              # program recompile: atoms re-translated, views rebuilt
        eng.upsert_throttles(base.throttle_batch([r0]), rows=np.array([r0], dtype=np.int32))
        sweep()
    finally:
        eng.close()


def _permute_pods(snap, rows):
    """A pods-only batch holding snap's pods in the order given by rows."""
    return _gather_pods([(snap, int(r)) for r in np.asarray(rows)])


def _gather_pods(entries):
    """A pods-only batch whose entry i is pod row r of snapshot s for entries[i] = (s, r) (same D / L everywhere)."""
    first = entries[0][0]
    b = S.Snapshot(first.D, first.L)
    b.alloc_namespaces(0, 0)
    b.alloc_throttles(0, 0, 0)
    nl = int(sum(s.pod_label_off[r + 1] - s.pod_label_off[r] for s, r in entries))
    nc = int(sum(s.pod_ctr_off[r + 1] - s.pod_ctr_off[r] for s, r in entries))
    b.alloc_pods(len(entries), nl, nc)
    lo = co = 0
    for i, (snap, r) in enumerate(entries):
        b.pod_ns[i], b.pod_flags[i] = snap.pod_ns[r], snap.pod_flags[r]
        l0, l1 = int(snap.pod_label_off[r]), int(snap.pod_label_off[r + 1])
        b.pod_label_key[lo:lo + l1 - l0] = snap.pod_label_key[l0:l1]
        b.pod_label_pair[lo:lo + l1 - l0] = snap.pod_label_pair[l0:l1]
        lo += l1 - l0
        b.pod_label_off[i + 1] = lo
        c0, c1 = int(snap.pod_ctr_off[r]), int(snap.pod_ctr_off[r + 1])
        b.ctr_init[co:co + c1 - c0] = snap.ctr_init[c0:c1]
        b.ctr_present[co:co + c1 - c0] = snap.ctr_present[c0:c1]
        b.ctr_req[co:co + c1 - c0] = snap.ctr_req[c0:c1]
        co += c1 - c0
        b.pod_ctr_off[i + 1] = co
        b.pod_ovh_present[i] = snap.pod_ovh_present[r]
        b.pod_ovh[i] = snap.pod_ovh[r]
    return b


def test_set_status_and_reserved(oracle_mod):
    """kt_set_status / kt_set_reserved feed what the informer cache / reserved cache hold."""
    snap = W.generate(W.small(seed=31, n_pods=800, n_thr=40, n_cluster=20))
    o = oracle_mod.Oracle(snap)
    rows = responsible_rows(snap)
    want = o.reconcile(NOW, rows=rows)
    eng = E.Engine.for_snapshot(snap)     # engine still has the EMPTY stored status
    try:
        snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod,
                          want.error, rows=rows)
        T = snap.n_thr
        allrows = np.arange(T, dtype=np.int32)
        eng.set_status(allrows, snap.thr_used, snap.thr_calc, (snap.thr_flags[:T] & S.THR_CALC_AT_NONZERO) != 0,
                       snap.thr_thrl_flag[:T], snap.thr_thrl_has[:T], (snap.thr_flags[:T] & S.THR_THROTTLED_POD) != 0,
                       snap.thr_status_msgs_fp[:T])
        # change some reservations on both sides
        res = S.Amounts(3, snap.D)
        for i in range(3):
            res.set_row(i, {0: 1000 * (i + 1), 1: 1 << 30}, count=i + 1)
            for f in ("v", "present", "count", "has_count"):
                getattr(snap.thr_reserved, f)[rows[i]] = getattr(res, f)[i]
        eng.set_reserved(rows[:3].astype(np.int32), res)
        st_w, sm_w = o.check()
        st_g, sm_g = eng.check(n=snap.n_pods, want_status=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()


@pytest.mark.parametrize("budget", [3000, 12000])
@pytest.mark.parametrize("seed", [1, 7])
def test_multi_chunk_index(seed, budget, oracle_mod, monkeypatch):
    """The index walked in several LDS-sized chunks (forced small here): terms of a throttle never straddle a chunk,
    class counters and errors carry from chunk to chunk, per-chunk `used` tables are summed per throttle."""
    monkeypatch.setenv("KT_CHUNK_BUDGET", str(budget))
    kw = dict(n_invalid_pod_sel=1, n_invalid_ns_sel=2, n_missing_ns=1) if seed == 7 else {}
    snap = W.generate(W.small(seed=seed, n_pods=3000, n_thr=192, n_cluster=96, **kw))
    st, sm, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    assert set(np.unique(st)) >= {S.NOT_AFFECTED, S.NOT_THROTTLED, S.ACTIVE}
    if seed == 7:
        assert (S.summary_fields(sm)[0] == S.VERDICT_ERROR).any() and rec.error.any()


def _spread_namespaces(snap, stride):
    """The same cluster with namespace row i moved to row i * stride (rows in between hold no Namespace object)."""
    import copy
    out = copy.copy(snap)
    n_new = (snap.n_ns - 1) * stride + 1
    out.n_ns = n_new
    out.ns_valid = np.zeros(n_new, dtype=np.uint8)
    out.ns_valid[::stride] = snap.ns_valid[:snap.n_ns]
    off = np.zeros(n_new + 1, dtype=np.uint32)
    counts = np.zeros(n_new, dtype=np.uint32)
    counts[::stride] = np.diff(snap.ns_label_off[:snap.n_ns + 1])
    off[1:] = np.cumsum(counts)
    out.ns_label_off = off          # the label arrays keep their order: only the rows' offsets move
    out.pod_ns = (snap.pod_ns.astype(np.uint64) * stride).astype(np.uint32)
    out.thr_ns = (snap.thr_ns.astype(np.uint64) * stride).astype(np.uint32)
    return out


@pytest.mark.parametrize("stride", [250, 600])
def test_namespace_order_with_many_namespace_rows(stride, oracle_mod, monkeypatch):
    """Namespace rows in use up to 5 751 / 13 801: the counting sort by namespace keeps a workgroup's counters in LDS
    while the rows fit (<= 12288 for the histogram, <= 4096 for the scatter) and falls back to global atomics beyond —
    both fallbacks, through a full parity run in namespace order (forced here: the program is small enough for one
    chunk).  The program covers the rows in USE, whatever the configured capacity."""
    monkeypatch.setenv("KT_FORCE_NS_ORDER", "1")

    def roomy(cls, snap, kernel_variant=E.VARIANT_INDEXED, device=-1, pod_capacity=None):
        e = cls(snap.D, max(snap.L, 1), pod_capacity or max(snap.n_pods, 1), max(snap.n_thr, 1), 100000, device, kernel_variant)
        e.load_snapshot(snap)
        return e

    monkeypatch.setattr(E.Engine, "for_snapshot", classmethod(roomy))
    snap = _spread_namespaces(W.generate(W.small(seed=15, n_pods=5000, n_thr=96, n_cluster=48, n_ns=24)), stride)
    st, sm, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    assert set(np.unique(st)) >= {S.NOT_AFFECTED, S.NOT_THROTTLED}


def test_pod_in_a_namespace_row_beyond_the_compiled_program(oracle_mod):
    """The program is compiled for the namespace rows in use; a pod that arrives later in a higher row (no Namespace
    object there yet) makes it recompile — its PreFilter is an Error (affectedClusterThrottles: Namespace lookup fails,
    clusterthrottle_controller.go:273-276), everything else is unchanged."""
    base = W.generate(W.small(seed=16, n_pods=600, n_thr=48, n_cluster=24, n_ns=6))
    eng = E.Engine(base.D, max(base.L, 1), 1000, max(base.n_thr, 1), 5000)
    try:
        eng.load_snapshot(base)
        st0, sm0 = eng.check(n=base.n_pods, want_status=False)
        stray = _permute_pods(base, np.array([0]))
        stray.pod_ns[0] = 4321
        eng.upsert_pods(stray, rows=np.array([600], dtype=np.int64))
        st1, sm1 = eng.check(n=601, want_status=False)
        np.testing.assert_array_equal(sm1[:600], sm0)
        assert S.summary_fields(sm1[600:601])[0][0] == S.VERDICT_ERROR
        eng.reconcile(NOW, apply=False)
    finally:
        eng.close()


def _stored_status(snap, oracle_mod):
    """Reconcile on the oracle and store the result as the snapshot's status (what UpdateStatus persists)."""
    o = oracle_mod.Oracle(snap)
    rows = responsible_rows(snap)
    want = o.reconcile(NOW, rows=rows)
    snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod,
                      want.error, rows=rows)


@pytest.mark.parametrize("hbm_state", [False, True], ids=["lds", "hbm"])
@pytest.mark.parametrize("seed,on_equal", [(52, False), (54, True), (53, False)])
def test_admit_queue_matches_sequential_prefilter_reserve(seed, on_equal, hbm_state, oracle_mod):
    """kt_admit_launch == for each pod in order: PreFilter, on Success Reserve; with the reserved
    amounts in LDS and (forced, as for thousands of throttles) in HBM."""
    if hbm_state:
        import subprocess, sys, os
        # the hook is read once per process: run this case in a child
        env = dict(os.environ, KT_ADMIT_FORCE_GLOBAL="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                            f"test_admit_queue_matches_sequential_prefilter_reserve and {seed}-{on_equal}-lds"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        return
    snap = W.generate(W.small(seed=seed, n_pods=4000, n_thr=64, n_cluster=32, n_invalid_pod_sel=1 if seed == 53 else 0,
                              n_missing_ns=1 if seed == 53 else 0))
    # head-room on every throttle (the generator calibrates a third of them as already throttled): the queue then
    # fills them up on the way
    T = snap.n_thr
    snap.thr_spec.v[:T] = snap.thr_spec.v[:T] * 2 + 1
    snap.thr_spec.count[:T] = snap.thr_spec.count[:T] * 2 + 3
    snap.thr_ovr_off[:] = 0  # no overrides: the spec threshold is the effective one
    _stored_status(snap, oracle_mod)
    o = oracle_mod.Oracle(snap)
    fl = snap.pod_flags[:snap.n_pods]
    pending = np.nonzero(((fl & S.POD_VALID) != 0) & ((fl & S.POD_SCHEDULED) == 0))[0]
    rng = np.random.default_rng(seed)
    queue = rng.permutation(pending)[:1500].astype(np.int64)
    st_w, sm_w, res_w = o.admit(queue, on_equal=on_equal)
    eng = E.Engine.for_snapshot(snap)
    try:
        # dry run first: same answers, reserved amounts untouched
        st_d, sm_d = eng.admit(queue, on_equal=on_equal, commit=False)
        np.testing.assert_array_equal(st_d, st_w)
        np.testing.assert_array_equal(sm_d, sm_w)
        res0 = eng.fetch_reserved()
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(res0, f)[:T], getattr(snap.thr_reserved, f)[:T], err_msg=f)
        # committed run
        st_g, sm_g = eng.admit(queue, on_equal=on_equal, commit=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
        res_g = eng.fetch_reserved()
        resp = responsible_rows(snap)
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(res_g, f)[resp], getattr(res_w, f)[resp], err_msg=f)
        # the reservations made on the way matter: a plain (unordered) check of the same pods answers differently
        st_c, sm_c = o.check(queue, on_equal=on_equal)
        assert (sm_w != sm_c).any()
        verdict = S.summary_fields(sm_w)[0]
        assert (verdict == S.VERDICT_ALLOW).any() and (verdict == S.VERDICT_BLOCK).any()
        # and the engine now checks against the advanced reserved amounts
        for f in ("v", "present", "count", "has_count"):
            getattr(snap.thr_reserved, f)[:T] = getattr(res_w, f)[:T]
        st2_w, sm2_w = o.check(queue[:500], on_equal=on_equal)
        st2_g, sm2_g = eng.check(queue[:500], on_equal=on_equal, want_status=True)
        np.testing.assert_array_equal(st2_g, st2_w)
        np.testing.assert_array_equal(sm2_g, sm2_w)
    finally:
        eng.close()


def test_admit_golden_prefix(oracle_mod):
    """Integration spec G1(ii)/G2(ii) as ONE queue (see tests/test_oracle_admit.py for the oracle side)."""
    from test_oracle_admit import build_prefix_case, EXPECTED_PREFIX
    built = build_prefix_case()
    snap = built.snapshot
    o = oracle_mod.Oracle(snap)
    st_w, sm_w, res_w = o.admit(None)
    eng = E.Engine.for_snapshot(snap)
    try:
        st_g, sm_g = eng.admit(n=snap.n_pods, commit=True)
        np.testing.assert_array_equal(st_g, st_w)
        np.testing.assert_array_equal(sm_g, sm_w)
        t_row, c_row = built.thr_names.index("default/t"), built.thr_names.index("/c")
        assert [(int(st_g[i, t_row]), int(st_g[i, c_row])) for i in range(6)] == EXPECTED_PREFIX
        res_g = eng.fetch_reserved()
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(res_g, f), getattr(res_w, f)[:snap.n_thr], err_msg=f)
    finally:
        eng.close()


def test_overflow_guard(oracle_mod):
    """resource.Quantity never overflows; the engine's exact range is int64.  A single request beyond 2^60 is refused at
    ingest; requests that only ADD UP beyond 2^60 are refused by the reconcile that would sum them (judged on their actual
    total, not on max x capacity); large requests whose total fits are summed exactly."""
    snap = W.generate(W.small(seed=41, n_pods=4, n_thr=2, n_cluster=0, D=2))
    snap.ctr_req[0, 0] = (1 << 60) + 1
    snap.ctr_present[0] |= 1
    eng = E.Engine(2, snap.L, 4, 2, snap.n_ns)
    with pytest.raises(E.EngineError) as ei:
        eng.load_snapshot(snap)
    assert ei.value.code == -4
    eng.close()
    # 64 pods x 2^55 = 2^61: every pod is fine, their sum leaves the range the int64 partials are proved for — an
    # incremental engine (it maintains int64 partials) refuses; a rescanning engine sums limbs (test_wide_sums)
    snap = W.generate(W.small(seed=41, n_pods=64, n_thr=4, n_cluster=2, D=2))
    nc = int(snap.pod_ctr_off[snap.n_pods])
    snap.ctr_req[:nc, 0] = 0
    snap.ctr_req[snap.pod_ctr_off[:snap.n_pods], 0] = 1 << 55   # first container of every pod
    snap.ctr_present[snap.pod_ctr_off[:snap.n_pods]] |= 1
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED | E.VARIANT_INCREMENTAL)
    with pytest.raises(E.EngineError) as ei:
        eng.reconcile(NOW, apply=False)
    assert ei.value.code == -4
    eng.close()
    # 64 pods x 2^53 = 2^59 fits: exact sums, equal to the oracle's 128-bit arithmetic
    snap.ctr_req[snap.pod_ctr_off[:snap.n_pods], 0] = 1 << 53
    eng = E.Engine.for_snapshot(snap)
    try:
        rows = responsible_rows(snap)
        want = oracle_mod.Oracle(snap).reconcile(NOW, rows=rows)
        got_all = eng.reconcile(NOW, apply=False)
        np.testing.assert_array_equal(got_all.used.v[rows], want.used.v[:len(rows)])
        assert int(got_all.used.v[rows].max()) >= 1 << 53
    finally:
        eng.close()


def _wide_parity(snap, oracle_mod, variant):
    """reconcile + check of a snapshot whose `used` sums leave int64, against the oracle's 128-bit arithmetic"""
    P, T, D = snap.n_pods, snap.n_thr, snap.D
    eng = E.Engine.for_snapshot(snap, variant)
    try:
        o = oracle_mod.Oracle(snap)
        rows = responsible_rows(snap)
        want = o.reconcile(NOW, rows=rows, nthreads=8, wide=True)
        assert not want.error[:len(rows)].any()
        got = eng.reconcile(NOW, apply=True)
        hi, any_wide = eng.reconcile_fetch_used_hi()
        assert any_wide and (want.used_hi[:len(rows)] != (want.used.v[:len(rows)] < 0) * -1).any(), "the case must leave int64"
        np.testing.assert_array_equal(got.used.v[rows], want.used.v[:len(rows)], err_msg="used, low words")
        np.testing.assert_array_equal(hi[rows], want.used_hi[:len(rows)], err_msg="used, high words")
        for f in ("present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(got.used, f)[rows], getattr(want.used, f)[:len(rows)], err_msg=f"used.{f}")
        np.testing.assert_array_equal(got.thrl_flag[rows], want.thrl_flag[:len(rows)])
        np.testing.assert_array_equal(got.thrl_pod[rows], want.thrl_pod[:len(rows)])
        # the check that follows reads the wide status: the oracle gets the same 128-bit values
        snap.apply_status(got.used, got.calc, got.calc_updated, got.thrl_flag, got.thrl_has, got.thrl_pod, got.error)
        o.set_status_used_hi(hi[:T])
        sample = np.unique(np.linspace(0, P - 1, min(P, 4096)).astype(np.int64))
        for on_equal in (False, True):
            st_w, sm_w = o.check(rows=sample, on_equal=on_equal, nthreads=8)
            st_g, sm_g = eng.check(rows=sample, on_equal=on_equal, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            np.testing.assert_array_equal(sm_g, sm_w)
        # a second reconcile (the stored wide status is its input) changes nothing
        again = eng.reconcile(NOW, apply=True)
        hi2, _ = eng.reconcile_fetch_used_hi()
        np.testing.assert_array_equal(again.used.v[:T], got.used.v[:T])
        np.testing.assert_array_equal(hi2, hi)
    finally:
        eng.close()


@pytest.mark.parametrize("variant", [E.VARIANT_INDEXED, E.VARIANT_DENSE], ids=["indexed", "dense"])
def test_wide_sums(variant, oracle_mod):
    """resource.Quantity never overflows (Add promotes to big decimals, pkg/resourcelist/resourcelist.go:48-54).  When the
    requests held add up beyond int64 the reconcile sums 32-bit limbs in two scans and kt_finalize joins them in 128 bits:
    `used` (low + high words), the throttled flags and every CheckThrottledFor status equal the oracle's __int128
    arithmetic.  64 pods x 2^59 of one resource, negative requests on another (the sums cancel below int64 again)."""
    snap = W.generate(W.small(seed=45, n_pods=64, n_thr=6, n_cluster=3, D=3))
    first = snap.pod_ctr_off[:snap.n_pods]
    nc = int(snap.pod_ctr_off[snap.n_pods])
    snap.ctr_req[:nc, 0] = 0
    snap.ctr_req[first, 0] = 1 << 59
    snap.ctr_req[first[::2], 1] = -(1 << 59)                 # every other pod: negative, the sums stay wide and signed
    snap.ctr_req[first[1::2], 1] = (1 << 59) + 12345
    snap.ctr_present[first] |= 3
    snap.thr_spec.v[:snap.n_thr, 0] = (1 << 62) + 7            # thresholds the wide sums are compared with
    snap.thr_spec.present[:snap.n_thr] |= 1
    _wide_parity(snap, oracle_mod, variant)


def test_wide_sums_ten_million_pods(oracle_mod):
    """SURVEY.md 7, hard part 1: 10^7 pods with byte-sized requests at milli scale.  8 Gi = 8.6e12 milli-bytes per pod; the
    ~1.5e6 pods a throttle counts add up to 1.3e19 > 2^63 — exact in the engine (limb sums) and in the oracle."""
    snap = W.generate(W.small(seed=44, n_pods=10_000_000, n_thr=6, n_cluster=3, D=2, n_ns=4, K=4, V=2, L=2, terms=(1, 1), reqs=(0, 1),
                              rich_ops=0, overrides=0))
    first = snap.pod_ctr_off[:snap.n_pods]
    snap.ctr_req[:, :] = 0
    snap.ctr_req[first, 1] = (8 << 30) * 1000
    snap.ctr_present[first] |= 2
    _wide_parity(snap, oracle_mod, E.VARIANT_INDEXED)


def test_api_errors():
    snap = W.generate(W.small(seed=42, n_pods=8, n_thr=2, n_cluster=1))
    eng = E.Engine.for_snapshot(snap)
    with pytest.raises(E.EngineError) as ei:
        eng.check_fetch(1)
    assert ei.value.code == -5
    with pytest.raises(E.EngineError) as ei:
        eng.check(rows=np.array([99], dtype=np.int64))
    assert ei.value.code == -2
    eng.close()
    # (a batch that names a row twice is legal since round 6 — the last entry wins: test_rows_named_twice_in_one_batch)


@pytest.mark.parametrize("budget", [None, 6000])
@pytest.mark.parametrize("shape", ["rich", "simple", "wide"])
def test_few_pod_checks(budget, shape, oracle_mod, monkeypatch):
    """kt_check with n <= 8 and no status matrix — one PreFilter call (plugin.go:148-215) — takes the few-pod path
    (kt_check_few: one wave per index chunk, summaries to pinned memory, shared engine lock).  Every summary word equals the
    oracle's, for both isThrottledOnEqual values, with one and with several index chunks, simple ({any}) and rich
    ({any, veto}, up to three positive keys) programs, a missing Namespace object (Error), and right after a reconcile was
    LAUNCHED (the check then reads the previous generation of CheckRecs or the new one: same pods, same status)."""
    if budget:
        monkeypatch.setenv("KT_CHUNK_BUDGET", str(budget))
    kw = {"rich": dict(seed=81, n_pods=3000, n_thr=96, n_cluster=48, n_missing_ns=1),
          "simple": dict(seed=82, n_pods=3000, n_thr=80, n_cluster=40, terms=(1, 1), reqs=(1, 2), rich_ops=0),
          "wide": dict(seed=83, n_pods=2000, n_thr=64, n_cluster=32, D=12, K=16, V=4, L=12)}[shape]
    snap = W.generate(W.small(**kw))
    eng = E.Engine.for_snapshot(snap)
    try:
        o = oracle_mod.Oracle(snap)
        got = eng.reconcile(NOW, apply=True)
        snap.apply_status(got.used, got.calc, got.calc_updated, got.thrl_flag, got.thrl_has, got.thrl_pod, got.error)
        o.refresh()
        rng = np.random.default_rng(81)
        for on_equal in (False, True):
            _, sm_w = o.check(want_status=False, on_equal=on_equal, nthreads=4)
            eng.check_atomic(rows=np.array([0], dtype=np.int64), on_equal=on_equal, want_status=False)  # records of this on_equal
            before = eng.few_checks_served()
            trials = 0
            for n in (1, 1, 1, 2, 3, 4, 5, 7, 8) * 6:
                rows = rng.integers(0, snap.n_pods, size=n).astype(np.int64)
                if trials % 9 == 4:
                    eng.reconcile_launch(NOW, True)  # kt_finalize in flight on the other stream while the check runs
                _, sm = eng.check_atomic(rows=rows, on_equal=on_equal, want_status=False)
                np.testing.assert_array_equal(sm, sm_w[rows], err_msg=f"n={n} rows={rows} on_equal={on_equal}")
                trials += 1
            assert eng.few_checks_served() - before == trials, "the few-pod path was not taken"
            # the staged path (status matrix requested) agrees
            rows = rng.integers(0, snap.n_pods, size=5).astype(np.int64)
            _, sm2 = eng.check_atomic(rows=rows, on_equal=on_equal, want_status=True)
            np.testing.assert_array_equal(sm2, sm_w[rows])
    finally:
        eng.close()


def test_throttle_events_that_leave_the_selectors_alone(oracle_mod):
    """A Throttle event whose selector terms, namespace and VALID / RESPONSIBLE / CLUSTER flags equal the stored row — a
    threshold edit, an override whose window moves, the controller's own status update coming back through the informer —
    only re-uploads the throttle tables: the compiled selector program, the index and the pods' atom rows stand
    (KT_COUNTER_COMPILES does not move).  Anything else recompiles.  Either way the next reconcile + check equal the
    oracle."""
    snap = W.generate(W.small(seed=91, n_pods=3000, n_thr=80, n_cluster=30, overrides=1))
    eng = E.Engine.for_snapshot(snap)
    try:
        def sweep():
            o = oracle_mod.Oracle(snap)
            rows = responsible_rows(snap)
            want = o.reconcile(NOW, rows=rows)
            got_all = eng.reconcile(NOW, apply=True)
            got = E.ReconcileResult(len(rows), snap.D)
            for name in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod", "error"):
                getattr(got, name)[:len(rows)] = getattr(got_all, name)[rows]
            for tab in ("used", "calc"):
                for f in ("v", "present", "count", "has_count"):
                    getattr(getattr(got, tab), f)[:len(rows)] = getattr(getattr(got_all, tab), f)[rows]
            assert_reconcile_equal(got, want, len(rows))
            snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows)
            for on_equal in (False, True):
                st_w, sm_w = o.check(on_equal=on_equal, nthreads=8)
                st_g, sm_g = eng.check(n=snap.n_pods, on_equal=on_equal, want_status=True)
                np.testing.assert_array_equal(st_g, st_w)
                np.testing.assert_array_equal(sm_g, sm_w)

        sweep()
        c0 = eng.compiles()
        assert c0 >= 1
        # threshold edits of four throttles, one of them with a different count threshold too
        rows = np.array([3, 17, 40, 61])
        snap.thr_spec.v[rows] = snap.thr_spec.v[rows] // 2 + 1
        snap.thr_spec.count[rows[:1]] = 2
        snap.thr_spec.has_count[rows[:1]] = 1
        eng.upsert_throttles(snap.throttle_batch(rows), rows=rows.astype(np.int32))
        sweep()
        assert eng.compiles() == c0, "a threshold edit recompiled the selector program"
        # an override window that starts later / ends earlier, a changed override threshold
        if snap.n_ovr:
            snap.ovr_begin_s[: snap.n_ovr : 2] += 7200
            snap.ovr_thr.v[: snap.n_ovr] = snap.ovr_thr.v[: snap.n_ovr] // 3 + 1
        eng.upsert_throttles(snap)           # every row again, status as the last sweep stored it
        sweep()
        assert eng.compiles() == c0, "re-feeding the throttles with unchanged selectors recompiled"
        # a single-pod PreFilter right after such an event takes the few-pod path again once the tables are up
        eng.check_atomic(rows=np.array([5], dtype=np.int64), want_status=False)
        # now events that DO change what a selector selects: a throttle stops being this scheduler's, a namespaced
        # Throttle moves to another namespace
        resp = responsible_rows(snap)
        r0 = int(resp[0])
        snap.thr_flags[r0] &= 0xFFFFFFFF ^ S.THR_RESPONSIBLE
        eng.upsert_throttles(snap.throttle_batch([r0]), rows=np.array([r0], dtype=np.int32))
        sweep()
        assert eng.compiles() == c0 + 1
        namespaced = [int(t) for t in resp[1:] if not (snap.thr_flags[t] & S.THR_CLUSTER)]
        r1 = namespaced[0]
        snap.thr_ns[r1] = (int(snap.thr_ns[r1]) + 1) % snap.n_ns
        eng.upsert_throttles(snap.throttle_batch([r1]), rows=np.array([r1], dtype=np.int32))
        sweep()
        assert eng.compiles() == c0 + 2
    finally:
        eng.close()


def test_partials_must_match_the_throttle_set(oracle_mod):
    """Between kt_aggregate_launch and the calls that consume its partials (the exchange, kt_finalize_launch) the throttle
    set must not change: a grown row count would read past what the scan filled and the ranks of an all-reduce would
    disagree on the word count.  The engine answers KT_ERR_NOT_READY and the caller aggregates again."""
    snap = W.generate(W.small(seed=43, n_pods=256, n_thr=6, n_cluster=3))
    more = W.generate(W.small(seed=43, n_pods=256, n_thr=9, n_cluster=4))
    eng = E.Engine(snap.D, max(snap.L, 1), 512, 16, max(snap.n_ns, more.n_ns, 1))
    try:
        eng.load_snapshot(snap)
        eng.aggregate_launch()
        eng.upsert_throttles(more)                      # 9 rows now: the pending partials hold 6
        with pytest.raises(E.EngineError) as ei:
            eng.finalize_launch(NOW, True)
        assert ei.value.code == -5
        eng.aggregate_launch()                          # again, against the new throttle set
        eng.finalize_launch(NOW, True)
        got = eng.reconcile_fetch()
        ref = E.Engine(more.D, max(more.L, 1), 512, 16, max(snap.n_ns, more.n_ns, 1))
        try:
            ref.upsert_namespaces(snap)
            ref.upsert_throttles(more)
            ref.upsert_pods(snap)
            want = ref.reconcile(NOW, apply=True)
        finally:
            ref.close()
        np.testing.assert_array_equal(got.used.v[:more.n_thr], want.used.v[:more.n_thr])
        np.testing.assert_array_equal(got.thrl_flag[:more.n_thr], want.thrl_flag[:more.n_thr])
        # a reconcile key that was never upserted is an error, not a silent "stored status unchanged"
        with pytest.raises(E.EngineError) as ei:
            eng.reconcile_rows(NOW, np.array([0, 12], dtype=np.int32))
        assert ei.value.code == -2
    finally:
        eng.close()


# ---------------------------------------------------------------------------------------------------
# BASELINE.json full-size configurations: parity on samples + size-independent properties
# ---------------------------------------------------------------------------------------------------
def _full_size_checks(cfg, oracle_mod, with_dense=True, nthreads=None, level="full", row_stride=10):
    """NOTHING is sampled: every responsible throttle's reconcile result and every pod's summary word are compared with
    the oracle (the C restatement runs the whole configuration in seconds on the GPU box's host cores), a pod sample
    additionally with full status rows, and the dense (reference-shaped) kernels must agree with the indexed ones."""
    nthreads = nthreads or oracle_mod.effective_cpus()  # (the cgroup quota, not the box's thread count)
    snap = W.generate(cfg)
    now = (cfg.now_s, 0)
    P, T = snap.n_pods, snap.n_thr
    eng = E.Engine.for_snapshot(snap)
    try:
        o = oracle_mod.Oracle(snap)
        # (1) reconcile: used / calculated threshold / throttled flags of EVERY responsible throttle, bit-exact
        #     (throttle_controller.go:116-133, clusterthrottle_controller.go:119-136)
        rows = responsible_rows(snap)
        want = o.reconcile(now, rows=rows, nthreads=nthreads)
        got = eng.reconcile(now, apply=True)
        assert not got.error[:T].any()
        if (got.used.v[rows] != want.used.v[:len(rows)]).any():  # say what differs before the assertion below fires
            g, w = got.used.v[rows], want.used.v[:len(rows)]
            bad = np.argwhere(g != w)
            print("used.v differs in", len(bad), "entries; per dimension", np.bincount(bad[:, 1], minlength=snap.D),
                  "; pod counts differ in", int((got.used.count[rows] != want.used.count[:len(rows)]).sum()), "throttles")
            for (i, d) in bad[:24]:
                print("   throttle", int(rows[i]), "cluster" if snap.thr_flags[rows[i]] & S.THR_CLUSTER else "namespaced", "dim", int(d), "got", int(g[i, d]),
                      "want", int(w[i, d]), "diff", int(g[i, d] - w[i, d]), "pods got / want", int(got.used.count[rows[i]]), int(want.used.count[i]))
            again = eng.reconcile(now, apply=False)
            print("   the same engine again:", int((again.used.v[rows] != w).sum()), "entries differ")
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(got.used, f)[rows], getattr(want.used, f)[:len(rows)], err_msg=f"used.{f}")
            np.testing.assert_array_equal(getattr(got.calc, f)[rows], getattr(want.calc, f)[:len(rows)], err_msg=f"calc.{f}")
        np.testing.assert_array_equal(got.thrl_flag[rows], want.thrl_flag[:len(rows)])
        np.testing.assert_array_equal(got.thrl_has[rows], want.thrl_has[:len(rows)])
        np.testing.assert_array_equal(got.thrl_pod[rows], want.thrl_pod[:len(rows)])
        np.testing.assert_array_equal(got.calc_updated[rows], want.calc_updated[:len(rows)])
        # (2) idempotence of reconcile
        again = eng.reconcile(now, apply=True)
        np.testing.assert_array_equal(again.used.v[:T], got.used.v[:T])
        assert not again.calc_updated[:T].any(), "second reconcile at the same instant must not replace thresholds"
        # (3) check: EVERY pod's summary word against the oracle on the engine's own status
        #     (throttle_controller.go:349-397, clusterthrottle_controller.go:378-425)
        snap.apply_status(got.used, got.calc, got.calc_updated, got.thrl_flag, got.thrl_has, got.thrl_pod, got.error)
        _, sm_w = o.check(want_status=False, nthreads=nthreads)
        _, sm_all = eng.check(n=P, want_status=False)
        np.testing.assert_array_equal(sm_all, sm_w)
        #     ... kt_sweep_launch: the same sweep AND a reconcile in one pass over the pod tables (the fused kernel for
        #     single-chunk programs, the pair of launches otherwise): every summary word and every `used` once more
        eng.sweep_launch(now, apply=True)
        _, sm_sw = eng.check_fetch(P, False)
        np.testing.assert_array_equal(sm_sw, sm_w, err_msg="kt_sweep_launch: summaries")
        rec_sw = eng.reconcile_fetch()
        for f in ("v", "present", "count", "has_count"):
            np.testing.assert_array_equal(getattr(rec_sw.used, f)[:T], getattr(got.used, f)[:T], err_msg=f"kt_sweep_launch: used.{f}")
        np.testing.assert_array_equal(rec_sw.thrl_flag[:T], got.thrl_flag[:T])
        np.testing.assert_array_equal(rec_sw.thrl_pod[:T], got.thrl_pod[:T])
        assert not rec_sw.calc_updated[:T].any()
        if level == "core":  # every `used`, every summary word, the sweep
            return sm_all
        #     ... and the full status ROWS — which throttle blocks the pod, not only how many of each kind (a summary word cannot
        #     tell a per-pair error that preserves the three counts): of EVERY pod where the matrix is a gigabyte (configs[2] /
        #     [3]: 10^9 cells, row_stride = 1), of every tenth pod on a configs[4] shard (1.25e10 cells); in slices of <= 2^27 cells.
        sample = np.arange(0, P, row_stride, dtype=np.int64)
        step = max(1, (1 << 27) // max(T, 1))
        for k0 in range(0, len(sample), step):
            part = sample[k0:k0 + step]
            st_w, sm_s = o.check(rows=part, nthreads=nthreads)
            st_g, sm_g = eng.check(rows=part, want_status=True)
            np.testing.assert_array_equal(st_g, st_w)
            np.testing.assert_array_equal(sm_g, sm_s)
        if level == "rows":  # ... which is where the other shards of a sharded configuration stop
            return sm_all
        # (4) every summary is self-consistent (verdict <=> some class count non-zero)
        verdict, n_exc, n_act, n_ins = S.summary_fields(sm_all)
        blocked = (n_exc + n_act + n_ins) > 0
        np.testing.assert_array_equal(verdict[verdict != S.VERDICT_ERROR] == S.VERDICT_BLOCK, blocked[verdict != S.VERDICT_ERROR])
        # (5) on_equal=True: every summary word against the oracle again (isThrottledOnEqual, plugin_args.go)
        _, sm_eq = eng.check(n=P, on_equal=True, want_status=False)
        _, sm_eq_w = o.check(want_status=False, on_equal=True, nthreads=nthreads)
        np.testing.assert_array_equal(sm_eq, sm_eq_w)
        assert (S.summary_fields(sm_eq)[0] >= verdict).all()
        if with_dense:
            dense = E.Engine.for_snapshot(snap, E.VARIANT_DENSE)
            try:
                _, sm_d = dense.check(n=P, want_status=False)
                np.testing.assert_array_equal(sm_d, sm_all)
                rd = dense.reconcile(now, apply=False)
                np.testing.assert_array_equal(rd.used.v[:T], got.used.v[:T])
                np.testing.assert_array_equal(rd.used.count[:T], got.used.count[:T])
                np.testing.assert_array_equal(rd.used.present[:T], got.used.present[:T])
            finally:
                dense.close()
        return sm_all
    finally:
        eng.close()


def test_config2_full_size(oracle_mod):
    """configs[2]: 1M pods x 1k Throttle+ClusterThrottle, D=8 — NOTHING sampled: ALL 10^6 summary words (both
    isThrottledOnEqual values), ALL throttles' `used` / thresholds / flags, and the full status row of EVERY pod (all 10^9
    (pod, throttle) cells, eight slices) against the oracle; the dense (reference-shaped) kernels agree with the indexed ones
    on all 10^9 decisions and every `used` vector."""
    _full_size_checks(W.preset(2), oracle_mod, row_stride=1)


def test_config3_overrides_full_size(oracle_mod):
    """configs[3]: same with temporaryThresholdOverrides active (time-window branch)."""
    cfg = W.preset(3)
    _full_size_checks(cfg, oracle_mod, row_stride=1)
    # the override branch really is exercised: thresholds differ from config 2's
    snap = W.generate(cfg)
    assert snap.n_ovr > 2 * snap.n_thr


@pytest.mark.parametrize("shard", range(8))
def test_config4_one_shard(oracle_mod, shard):
    """configs[4]: 10M pods x 10k throttles with multi-term OR-of-AND selectors — the rows of EVERY 1/8 shard (the per-GPU
    slices of the 8-GPU configuration), NOTHING sampled: all 1.25M summary words (1.25e10 decisions) and all 10k throttles'
    `used` of each shard against the oracle, kt_sweep_launch once more, and every tenth pod with its full status row (round 6: on
    every shard).  Shards 0, 3 and 7 also compare every summary word under isThrottledOnEqual; the dense kernels (1.25e10 pair
    evaluations in the reference loop shape) cross-check shard 3.  (Round 5: the oracle's test mode evaluates the namespace side of a
    ClusterThrottle term once per (term, namespace) — kto_enable_ns_memo — which is what made all eight affordable.)"""
    cfg = W.preset(4).shard(shard, 8)
    _full_size_checks(cfg, oracle_mod, with_dense=(shard == 3), level="full" if shard in (0, 3, 7) else "rows")


@pytest.mark.parametrize("n_terms", [65, 130])
def test_throttles_with_more_than_64_terms(n_terms, oracle_mod):
    """The reference takes any number of selector terms (throttle_selector.go:30-42).  A throttle with more than 64 terms is ONE
    run of term numbers across several 64-bit words of the index (round 6; the slow list until round 5: 500x the step of the same
    terms in the index): the scans that dedupe match by match take such a program — the full check, the plain fold — and "reported
    once" holds across the words.  Every throttle of this cluster has 65 / 130 terms; everything is compared with the oracle, and
    NO throttle is on the slow list (KT_COUNTER_SLOW_THROTTLES)."""
    cfg = W.small(seed=700 + n_terms, n_pods=3000, n_thr=12, n_cluster=6, K=16, V=8, L=6, terms=(n_terms, n_terms), reqs=(1, 3))
    snap = W.generate(cfg)
    assert int(np.diff(snap.thr_term_off[:snap.n_thr + 1]).min()) >= n_terms
    st, sm, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    assert (st != S.NOT_AFFECTED).any()
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
    try:
        eng.reconcile(NOW, apply=False)
        assert eng.index_stats()["slow_throttles"] == 0
        # one PreFilter call on such a program: not the few-pod kernel (lane = (pod, word) cannot dedupe across words), same answer
        o = oracle_mod.Oracle(snap)
        _, sm_w = o.check(rows=np.arange(8, dtype=np.int64), want_status=False)
        _, sm_g = eng.check(rows=np.arange(8, dtype=np.int64), want_status=False)
        np.testing.assert_array_equal(sm_g, sm_w)
    finally:
        eng.close()
    # ... and beside ordinary throttles (the slow list next to indexed ones, several chunks)
    mixed = W.generate(W.small(seed=710 + n_terms, n_pods=3000, n_thr=48, n_cluster=24, K=16, V=8, L=6, terms=(1, n_terms), reqs=(1, 3)))
    assert int(np.diff(mixed.thr_term_off[:mixed.n_thr + 1]).max()) > 64
    run_full_parity(mixed, oracle_mod, E.VARIANT_INDEXED)


def test_terms_with_four_and_five_positive_keys_100k(oracle_mod):
    """A term with four or five positive keys is decided by the bitmaps alone since round 6 — the NEED = 5 instantiations count
    hits per term as 3-bit numbers (until round 5 such a term kept ONE anchor and every candidate went through the generic walk:
    13.6x the step of the same cluster with <= 3 keys); only a term with MORE than five keeps five and is confirmed by the walk
    (`slow` in the word header, confirm() in kt_scan.h).  100k pods against throttles whose terms carry
    three to five requirements: at least a tenth of the terms have four or five POSITIVE keys; everything against the
    oracle (status matrix included)."""
    cfg = W.small(seed=4545, n_pods=100000, n_thr=160, n_cluster=80, n_ns=16, K=16, V=3, L=10, terms=(1, 3), reqs=(3, 5))
    snap = W.generate(cfg)
    n_terms = int(snap.thr_term_off[snap.n_thr])
    many = 0
    for g in range(n_terms):
        ops = snap.preq.op[snap.term_preq_off[g]:snap.term_preq_off[g + 1]]
        many += int(((ops == 0) | (ops == 2)).sum() >= 4)  # In / Exists (distinct keys inside a term: the generator's rule)
    assert many * 10 >= n_terms, f"{many} of {n_terms} terms with >= 4 positive keys"
    st, sm, rec = run_full_parity(snap, oracle_mod, E.VARIANT_INDEXED)
    assert (st == S.NOT_THROTTLED).any() or (st == S.ACTIVE).any()


def test_sixteen_dims_sixteen_labels_1m(oracle_mod):
    """D = 16 resource names and L = 16 labels per pod are a product path (resourcelist.go:27-54: any number of names), not a
    3000-pod shape: configs[2]'s cluster at full size — 1M pods x 1k throttles — with 16 dimensions and 16 labels over 32
    keys (the <16, 16, rich> instantiations of both scans): every `used`, every summary word (both isThrottledOnEqual values),
    every tenth pod with its full status row."""
    cfg = W.preset(2)
    cfg.D, cfg.L, cfg.K = 16, 16, 32
    cfg.terms_min, cfg.terms_max, cfg.reqs_min, cfg.reqs_max, cfg.rich_ops = 1, 2, 1, 3, 1
    _full_size_checks(cfg, oracle_mod, with_dense=False)


# ---------------------------------------------------------------------------------------------------
# The fused packed reconcile reads every launched aggregate workgroup's slab (round 3: a workgroup without tiles
# returned before spilling, so its slab held whatever the allocator or an earlier, larger scan of the same engine left)
# ---------------------------------------------------------------------------------------------------
def _countable(flags):
    need = S.POD_VALID | S.POD_SCHED_MATCH | S.POD_SCHEDULED
    return (flags & need) == need


def test_reconcile_after_a_larger_scan_of_the_same_engine(oracle_mod):
    """Deterministic regression for VERDICT r3 weak #1.  ONE engine: first a snapshot of configs[2] with 655 360
    countable pods (10 240 tiles = 40 per workgroup: all 256 aggregate workgroups busy, every slab holds sums), then
    plain configs[2] (650 374 countable pods: 10 163 tiles, workgroup 255 owns none; the slab area is kept) -> the
    reconcile must equal the oracle.  On the round-3 library the stale slab of workgroup 255 is summed again."""
    cfg = W.preset(2)
    snap = W.generate(cfg)
    now = (cfg.now_s, 0)
    flags0 = snap.pod_flags[:snap.n_pods].copy()
    n_c = int(_countable(flags0).sum())
    tiles = (n_c + 63) // 64
    tpb = (tiles + 255) // 256
    assert (tiles + tpb - 1) // tpb < 256, "premise: plain configs[2] leaves an aggregate workgroup without tiles"
    target = 256 * tpb * 64
    extra = np.nonzero(~_countable(flags0) & ((flags0 & S.POD_VALID) != 0))[0][:target - n_c]
    assert len(extra) == target - n_c
    rows = responsible_rows(snap)
    eng = E.Engine.for_snapshot(snap)
    try:
        snap.pod_flags[extra] |= S.POD_SCHED_MATCH | S.POD_SCHEDULED
        eng.load_snapshot(snap)
        big = eng.reconcile(now, apply=False)
        snap.pod_flags[:snap.n_pods] = flags0
        eng.load_snapshot(snap)
        got = eng.reconcile(now, apply=False)
        want = oracle_mod.Oracle(snap).reconcile(now, rows=rows, nthreads=oracle_mod.effective_cpus())
        assert (big.used.count[rows] >= want.used.count[:len(rows)]).all() and (big.used.count[rows] > want.used.count[:len(rows)]).any()
        bad = np.argwhere(got.used.v[rows] != want.used.v[:len(rows)])
        assert len(bad) == 0, "used.v differs in %d entries of %d throttles (kernels: %s, %s)" % (
            len(bad), len(np.unique(bad[:, 0])), eng.kernel_name(E.KERNEL_AGGREGATE), eng.kernel_name(E.KERNEL_FINALIZE))
        np.testing.assert_array_equal(got.used.count[rows], want.used.count[:len(rows)])
        np.testing.assert_array_equal(got.used.present[rows], want.used.present[:len(rows)])
        np.testing.assert_array_equal(got.thrl_flag[rows], want.thrl_flag[:len(rows)])
    finally:
        snap.pod_flags[:snap.n_pods] = flags0
        eng.close()


@pytest.mark.parametrize("preset", [2, 3])
def test_reconcile_stress_fresh_engines(oracle_mod, preset):
    """configs[2] / [3]: KT_STRESS_ROUNDS (default 60; profiles/r05_stress.log: 300) reconciles, each on a FRESH engine,
    against ONE oracle result — between rounds device memory of many sizes is filled with 0xFF and freed, so that a read
    of memory no launch wrote cannot hide behind an allocator that hands back zeroed pages.  Every fifth engine
    reconciles three times more (the meeting of throttles with several groups runs again on warm caches)."""
    import ctypes
    rounds = int(os.environ.get("KT_STRESS_ROUNDS", "60"))
    cfg = W.preset(preset)
    snap = W.generate(cfg)
    now = (cfg.now_s, 0)
    rows = responsible_rows(snap)
    want = oracle_mod.Oracle(snap).reconcile(now, rows=rows, nthreads=oracle_mod.effective_cpus())
    hip = ctypes.CDLL("libamdhip64.so")
    rng = np.random.default_rng(11 + preset)

    def garbage():
        ptrs = []
        for sz in list(rng.integers(1 << 12, 1 << 23, size=24)) + [40 << 20, 64 << 20, 12 << 20]:
            q = ctypes.c_void_p()
            if hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(int(sz))) == 0:
                hip.hipMemset(q, 0xFF, ctypes.c_size_t(int(sz)))
                ptrs.append(q)
        hip.hipDeviceSynchronize()
        for q in ptrs:
            hip.hipFree(q)

    failures = []
    for rep in range(rounds):
        garbage()
        eng = E.Engine.for_snapshot(snap)
        try:
            for k in range(4 if rep % 5 == 0 else 1):
                got = eng.reconcile(now, apply=False)
                bad = np.argwhere(got.used.v[rows] != want.used.v[:len(rows)])
                nc = int((got.used.count[rows] != want.used.count[:len(rows)]).sum())
                nf = int((got.thrl_flag[rows] != want.thrl_flag[:len(rows)]).sum())
                if len(bad) or nc or nf:
                    d = (got.used.v[rows] - want.used.v[:len(rows)])[got.used.v[rows] != want.used.v[:len(rows)]]
                    failures.append("round %d.%d: used.v differs in %d entries (%d throttles, dims %s, |diff| <= %d), counts in %d, flags in %d" % (
                        rep, k, len(bad), len(np.unique(bad[:, 0])) if len(bad) else 0,
                        np.bincount(bad[:, 1], minlength=snap.D).tolist() if len(bad) else [], int(np.abs(d).max()) if len(d) else 0, nc, nf))
        finally:
            eng.close()
    print("configs[%d]: %d fresh-engine rounds, %d mismatching reconciles" % (preset, rounds, len(failures)))
    assert not failures, "\n".join(failures[:20])


@pytest.mark.parametrize("budget", [None, 6000])
def test_event_bursts_then_prefilter(budget, oracle_mod, monkeypatch):
    """Pod feed calls return once their kernels are enqueued (round 4: pinned event slots, no stream synchronisation) and
    pipeline among themselves; every other entry point waits for them first.  Bursts of 1-20 single-pod upserts / deletes
    (more than the eight slots: slots are reused), each followed AT ONCE by kt_check(n = 1..4) of pods the burst touched
    — the few-pod path on its own stream — and now and then by a reconcile: every summary word and every `used` equals the
    oracle on the pods as fed (throttle_controller.go:400-536: the handler's effect is visible to the next PreFilter)."""
    if budget:
        monkeypatch.setenv("KT_CHUNK_BUDGET", str(budget))
    base = W.generate(W.small(seed=74, n_pods=1800, n_thr=80, n_cluster=40))
    P = 1500
    rng = np.random.default_rng(74)
    state = np.full(P, -1, dtype=np.int64)
    state[:1000] = np.arange(1000)
    eng = E.Engine(base.D, max(base.L, 1), P, max(base.n_thr, 1), max(base.n_ns, 1))
    try:
        eng.upsert_namespaces(base)
        eng.upsert_throttles(base)
        eng.upsert_pods(_permute_pods(base, np.arange(1000)), rows=np.arange(1000))
        rows_t = responsible_rows(base)

        def reconcile_and_compare():
            n = int(np.nonzero(state >= 0)[0].max()) + 1
            snap = _with_pods(base, state[:n])
            want = oracle_mod.Oracle(snap).reconcile(NOW, rows=rows_t)
            got = eng.reconcile(NOW, apply=True)
            ok = want.error[:len(rows_t)] == 0
            np.testing.assert_array_equal(got.used.v[rows_t][ok], want.used.v[:len(rows_t)][ok])
            np.testing.assert_array_equal(got.used.count[rows_t][ok], want.used.count[:len(rows_t)][ok])
            snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows_t)
            base.thr_used, base.thr_calc = snap.thr_used, snap.thr_calc
            base.thr_flags, base.thr_thrl_flag, base.thr_thrl_has = snap.thr_flags, snap.thr_thrl_flag, snap.thr_thrl_has

        reconcile_and_compare()
        checkers = {k: eng.checker(k) for k in (1, 2, 3, 4)}
        checkers[1][2]()                                                 # CheckRecs of on_equal = False are built
        served0 = eng.few_checks_served()
        n_checks = 0
        for burst in range(40):
            touched = []
            for _ in range(int(rng.integers(1, 21))):
                r = int(rng.integers(0, P))
                if rng.random() < .2 and state[r] >= 0:
                    state[r] = -1
                    eng.delete_pods(np.array([r], dtype=np.int64))
                else:
                    state[r] = int(rng.integers(0, 1800))
                    eng.upsert_pods(_permute_pods(base, state[[r]]), rows=np.array([r]))
                    touched.append(r)
            live = [r for r in dict.fromkeys(touched) if state[r] >= 0]
            if live:
                pick = np.array(live[-int(rng.integers(1, 5)):], dtype=np.int64)
                rows_c, sum_c, call_c = checkers[len(pick)]
                rows_c[:] = pick
                call_c()                                                 # kt_check right behind the burst: nothing in between
                sm = sum_c.copy()
                n_checks += 1
                n = int(np.nonzero(state >= 0)[0].max()) + 1
                snap = _with_pods(base, state[:n])
                _, sm_w = oracle_mod.Oracle(snap).check(rows=pick, want_status=False, nthreads=2)
                np.testing.assert_array_equal(sm, sm_w, err_msg=f"burst {burst}: rows {pick}")
            if burst % 8 == 7:
                reconcile_and_compare()
        assert eng.few_checks_served() - served0 >= n_checks - 5, "the few-pod path was hardly taken"
    finally:
        eng.close()
