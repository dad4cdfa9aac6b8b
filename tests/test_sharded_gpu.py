"""The multi-GPU result, pinned on ONE GPU: the pod rows of a snapshot are cut into `world` contiguous shards exactly as
bench.py --gpus N cuts them, every shard runs through its own engine (kt_aggregate_launch into a caller
buffer), the partial-`used` buffers are summed the way the RCCL all-reduce sums them, the sum is handed back to every
shard's engine (kt_use_partial_buffer + kt_finalize_launch, APPLY) and each shard's PreFilter sweep follows.  Everything
that comes out — `used`, calculated thresholds, throttled flags of EVERY throttle, the summary word of EVERY pod of every
shard and the full status rows of a pod sample — must equal the CPU oracle's answer on the UNSHARDED snapshot, bit for bit
(throttle_controller.go:116-133 semantics: one reconcile over all pods).  Integer sums are associative, so this is the
result any number of GPUs produces; what an 8-GPU node adds is only the transport of the sum."""
import numpy as np
import pytest

from kube_throttler_amd import engine as E
from kube_throttler_amd import snapshot as S
from kube_throttler_amd import workload as W

pytestmark = pytest.mark.gpu


def _responsible(snap):
    need = S.THR_VALID | S.THR_RESPONSIBLE
    return np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]


def _sharded_pipeline(full_cfg, world, oracle_mod, n_thr_sample, n_pod_sample, nthreads=32):
    import torch
    nthreads = nthreads or oracle_mod.effective_cpus()
    full = W.generate(full_cfg)
    now = (full_cfg.now_s, 0)
    T, D = full.n_thr, full.D
    words = T * E.partial_layout(D)["stride"]  # (the library's own statement of the row layout: kt_partial_layout)
    # ---- pass 1: every shard's partial buffer; the host-side sum stands in for the all-reduce
    total = torch.zeros(words, dtype=torch.int64, device="cuda")
    shard_rows = []
    for r in range(world):
        cfg = full_cfg.shard(r, world)
        shard_rows.append((int(cfg.pod_begin), int(cfg.n_pods)))
        snap = W.generate(cfg)
        eng = E.Engine.for_snapshot(snap)
        try:
            assert eng.partial_words() == words
            part = torch.zeros(words, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.use_partial_buffer(part.data_ptr(), words)
            eng.aggregate_launch()
            eng.synchronize()
            total += part
            torch.cuda.synchronize()
            eng.use_partial_buffer(None, 0)
        finally:
            eng.close()
    assert sum(n for _, n in shard_rows) == full.n_pods
    # ---- the oracle on the unsharded snapshot
    o = oracle_mod.Oracle(full)
    rows = _responsible(full)
    pick = rows[np.unique(np.linspace(0, len(rows) - 1, min(n_thr_sample, len(rows))).astype(int))]
    want = o.reconcile(now, rows=pick, nthreads=nthreads)
    # ---- pass 2: the sum goes back to every shard: finalize (APPLY) + PreFilter sweep of the shard's pods
    first = None
    for r in range(world):
        snap = W.generate(full_cfg.shard(r, world))
        eng = E.Engine.for_snapshot(snap)
        try:
            buf = total.clone()
            torch.cuda.synchronize()
            eng.use_partial_buffer(buf.data_ptr(), words)
            eng.finalize_launch(now, apply=True)
            got = eng.reconcile_fetch()
            assert not got.error[:T].any()
            for f in ("v", "present", "count", "has_count"):
                np.testing.assert_array_equal(getattr(got.used, f)[pick], getattr(want.used, f)[:len(pick)], err_msg=f"shard {r} used.{f}")
                np.testing.assert_array_equal(getattr(got.calc, f)[pick], getattr(want.calc, f)[:len(pick)], err_msg=f"shard {r} calc.{f}")
            np.testing.assert_array_equal(got.thrl_flag[pick], want.thrl_flag[:len(pick)], err_msg=f"shard {r}")
            np.testing.assert_array_equal(got.thrl_pod[pick], want.thrl_pod[:len(pick)], err_msg=f"shard {r}")
            if first is None:
                first = got
                # the status every replica now holds, for the oracle's PreFilter
                full.apply_status(got.used, got.calc, got.calc_updated, got.thrl_flag, got.thrl_has, got.thrl_pod, got.error)
            else:  # every replica finalizes the same sums: identical results on every rank
                for f in ("v", "present", "count"):
                    np.testing.assert_array_equal(getattr(got.used, f)[:T], getattr(first.used, f)[:T])
                np.testing.assert_array_equal(got.thrl_flag[:T], first.thrl_flag[:T])
                np.testing.assert_array_equal(got.calc_updated[:T], first.calc_updated[:T])
            begin, n = shard_rows[r]
            local = np.unique(np.linspace(0, n - 1, min(n_pod_sample, n)).astype(np.int64))
            st_w, sm_w = o.check(rows=local + begin, nthreads=nthreads)
            st_g, sm_g = eng.check(rows=local, want_status=True)
            np.testing.assert_array_equal(st_g, st_w, err_msg=f"shard {r} status rows")
            np.testing.assert_array_equal(sm_g, sm_w, err_msg=f"shard {r} summaries")
            # the PreFilter sweep of the shard: EVERY pod's summary word
            _, sm_all = eng.check(n=n, want_status=False)
            _, sm_all_w = o.check(rows=np.arange(begin, begin + n, dtype=np.int64), want_status=False, nthreads=nthreads)
            np.testing.assert_array_equal(sm_all, sm_all_w, err_msg=f"shard {r} sweep")
            eng.use_partial_buffer(None, 0)
        finally:
            eng.close()


def test_eight_shards_of_a_multi_term_program(oracle_mod):
    """configs[4] scaled down (80k pods x 2k throttles, 2-4 terms per throttle, every selector operator, 256
    namespaces): 8 shards, every throttle, every pod's summary word and the status rows of 2048 pods per shard against the oracle."""
    cfg = W.preset(4)
    cfg.n_pods_total = cfg.n_pods = 80000
    cfg.n_thr, cfg.n_cluster = 2000, 1000
    _sharded_pipeline(cfg, 8, oracle_mod, n_thr_sample=2000, n_pod_sample=2048)


def test_eight_shards_of_config2(oracle_mod):
    """configs[2] at full size (1M pods x 1k throttles) in 8 shards of 125k pods, nothing sampled where it counts: ALL 1000
    throttles' `used` / thresholds / flags and EVERY pod's summary word of every shard against the oracle on the unsharded
    snapshot (plus the full status rows of 8192 pods per shard)."""
    _sharded_pipeline(W.preset(2), 8, oracle_mod, n_thr_sample=1000, n_pod_sample=8192, nthreads=None)


def test_partial_buffer_is_laid_out_as_the_query_says(oracle_mod):
    """What kt_aggregate_launch leaves in the partial buffer, decoded by tests/partial_layout.py — whose stride and offsets
    come from kt_partial_layout, the query the gloo world-size-2 test (tests/test_distributed_cpu.py) lays its buffers out
    by — must be the oracle's `used` of the same pods; and the other way round: the oracle's `used` packed by the same helper
    and handed to kt_finalize_launch gives the oracle's reconcile result.  Both kernel variants, 8 and 12 dimensions."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import partial_layout as KD
    for D, variant in ((8, E.VARIANT_INDEXED), (12, E.VARIANT_INDEXED), (8, E.VARIANT_DENSE)):
        cfg = W.small(seed=90 + D, n_pods=3000, n_thr=64, n_cluster=32, D=D, n_invalid_pod_sel=2, n_missing_ns=1)
        snap = W.generate(cfg)
        now = (cfg.now_s, 0)
        T = snap.n_thr
        lo = E.partial_layout(D)
        assert lo["stride"] >= 2 * D + 2 and len({lo["values"], lo["presence"], lo["pods"], lo["errors"]}) == 4
        want = oracle_mod.Oracle(snap).reconcile(now)
        eng = E.Engine.for_snapshot(snap, variant)
        try:
            words = eng.partial_words()
            assert words == T * lo["stride"]
            part = torch.zeros(words, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.use_partial_buffer(part.data_ptr(), words)
            eng.aggregate_launch()
            eng.synchronize()
            v, present, count, has_count, err = KD.unpack_partial(part.cpu().numpy(), D)
            rows = _responsible(snap)
            ok = rows[want.error[rows] == 0]
            np.testing.assert_array_equal(err[rows], want.error[rows] != 0)
            np.testing.assert_array_equal(v[ok], want.used.v[ok])
            np.testing.assert_array_equal(present[ok], want.used.present[ok])
            np.testing.assert_array_equal(count[ok], want.used.count[ok])
            # the oracle's `used`, packed by the helper, through the engine's finalize
            packed = torch.from_numpy(KD.pack_partial(want.used.v, want.used.present, want.used.count, want.error, D).reshape(-1).copy()).cuda()
            torch.cuda.synchronize()
            eng.use_partial_buffer(packed.data_ptr(), words)
            eng.finalize_launch(now, apply=False)
            got = eng.reconcile_fetch()
            np.testing.assert_array_equal(got.used.v[ok], want.used.v[ok])
            np.testing.assert_array_equal(got.used.present[ok], want.used.present[ok])
            np.testing.assert_array_equal(got.thrl_flag[ok], want.thrl_flag[ok])
            np.testing.assert_array_equal(got.thrl_pod[ok], want.thrl_pod[ok])
            eng.use_partial_buffer(None, 0)
        finally:
            eng.close()


def test_three_uneven_shards_with_selector_errors(oracle_mod):
    """Shards of different length (2001 pods over 3 ranks), unconvertible selectors and missing Namespace objects:
    the error words of the partial buffer travel through the sum as well."""
    cfg = W.small(seed=21, n_pods=2001, n_thr=96, n_cluster=48, n_invalid_pod_sel=3, n_invalid_ns_sel=2, n_missing_ns=1)
    import torch
    full = W.generate(cfg)
    now = (cfg.now_s, 0)
    T, D = full.n_thr, full.D
    words = T * E.partial_layout(D)["stride"]  # (the library's own statement of the row layout: kt_partial_layout)
    total = torch.zeros(words, dtype=torch.int64, device="cuda")
    for r in range(3):
        eng = E.Engine.for_snapshot(W.generate(cfg.shard(r, 3)))
        try:
            part = torch.zeros(words, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.use_partial_buffer(part.data_ptr(), words)
            eng.aggregate_launch()
            eng.synchronize()
            total += part
            torch.cuda.synchronize()
            eng.use_partial_buffer(None, 0)
        finally:
            eng.close()
    want = oracle_mod.Oracle(full).reconcile(now)
    eng = E.Engine.for_snapshot(W.generate(cfg.shard(0, 3)))
    try:
        eng.use_partial_buffer(total.data_ptr(), words)
        eng.finalize_launch(now, apply=False)
        got = eng.reconcile_fetch()
        rows = _responsible(full)
        np.testing.assert_array_equal(got.error[rows] != 0, want.error[rows] != 0)
        assert want.error[rows].any()
        ok = rows[want.error[rows] == 0]
        np.testing.assert_array_equal(got.used.v[ok], want.used.v[ok])
        np.testing.assert_array_equal(got.used.count[ok], want.used.count[ok])
        np.testing.assert_array_equal(got.thrl_flag[ok], want.thrl_flag[ok])
        eng.use_partial_buffer(None, 0)
    finally:
        eng.close()


# ---------------------------------------------------------------------------------------------------
# Two real GPUs, one process each, the engine's own RCCL exchange between them (skipped on a one-GPU box)
# ---------------------------------------------------------------------------------------------------
def _rank_main(rank, world, uid, cfg_bytes, out_q):
    """One rank: its shard on device `rank`, kt_aggregate_launch -> kt_comm_allreduce_partial -> kt_finalize_launch
    (APPLY) -> the PreFilter sweep of its pods; results go back to the parent for the comparison."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from kube_throttler_amd import engine as E2, workload as W2
    full_cfg = W2.WorkloadCfg.from_buffer_copy(cfg_bytes)
    snap = W2.generate(full_cfg.shard(rank, world))
    eng = E2.Engine.for_snapshot(snap, device=rank)
    try:
        eng.comm_init(rank, world, uid)
        eng.aggregate_launch()
        eng.comm_allreduce_partial()
        eng.finalize_launch((full_cfg.now_s, 0), True)
        rec = eng.reconcile_fetch()
        st, sm = eng.check(n=snap.n_pods, on_equal=False, want_status=True)
        eng.comm_destroy()
        out_q.put((rank, rec.used.v.copy(), rec.used.present.copy(), rec.used.count.copy(), rec.thrl_flag.copy(), st, sm))
    finally:
        eng.close()


def test_two_gpus_with_the_native_exchange(oracle_mod):
    """world_size 2 on real hardware: pods row-sharded over two GPUs, partials summed by ncclAllReduce(int64, sum)
    inside the engines (kt_comm_*), finalize replicated, check local — against the oracle on the unsharded snapshot."""
    import multiprocessing as mp
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the one-GPU form of this check is test_eight_shards_*)")
    full_cfg = W.small(seed=91, n_pods=6000, n_thr=96, n_cluster=48)
    full = W.generate(full_cfg)
    now = (full_cfg.now_s, 0)
    uid = E.Engine.comm_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, uid, bytes(full_cfg), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            item = q.get(timeout=300)
            got[item[0]] = item[1:]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:  # a rank that died leaves its peer inside the collective: do not leave it behind
            if p.is_alive():
                p.kill()
    o = oracle_mod.Oracle(full)
    rows = _responsible(full)
    want = o.reconcile(now, rows=rows, nthreads=8)
    for r in range(2):  # finalize ran replicated: both ranks hold the whole result
        used_v, used_p, used_c, thrl, _, _ = got[r]
        np.testing.assert_array_equal(used_v[rows], want.used.v[:len(rows)])
        np.testing.assert_array_equal(used_p[rows], want.used.present[:len(rows)])
        np.testing.assert_array_equal(used_c[rows], want.used.count[:len(rows)])
        np.testing.assert_array_equal(thrl[rows], want.thrl_flag[:len(rows)])
    full.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows)
    st_w, sm_w = o.check(on_equal=False, nthreads=8)
    begin = 0
    for r in range(2):
        st, sm = got[r][4], got[r][5]
        np.testing.assert_array_equal(st, st_w[begin:begin + len(sm)])
        np.testing.assert_array_equal(sm, sm_w[begin:begin + len(sm)])
        begin += len(sm)
    assert begin == full.n_pods


def test_two_skewed_ranks_agree_on_wide_sums(oracle_mod):
    """whether a rank's requests leave int64 is a LOCAL fact.  Rank 0 holds 64 pods x 2^59 (its sums leave
    int64), rank 1 holds 64 small pods (its sums do not).  Left to themselves the ranks would exchange buffers of
    different layouts; so with exchange_world = 2 the wide rank refuses (KT_ERR_OVERFLOW_RISK, as rounds 1-2 did) until the
    host switches EVERY rank with kt_set_wide_sums(1) — then both fill two blocks, the blocks add up rank-wise, and the
    finalize on the sum equals the oracle's 128-bit arithmetic on the unsharded snapshot."""
    import torch
    cfg = W.small(seed=46, n_pods=128, n_thr=6, n_cluster=3, D=3)

    def shaped(c, row0):
        snap = W.generate(c)
        first = snap.pod_ctr_off[:snap.n_pods]
        nc = int(snap.pod_ctr_off[snap.n_pods])
        snap.ctr_req[:nc, 0] = 0
        big = (np.arange(snap.n_pods) + row0) < 64
        snap.ctr_req[first[big], 0] = 1 << 59
        snap.ctr_req[first[~big], 0] = 1000
        snap.ctr_present[first] |= 1
        snap.thr_spec.v[:snap.n_thr, 0] = (1 << 62) + 7
        snap.thr_spec.present[:snap.n_thr] |= 1
        return snap

    full = shaped(cfg, 0)
    now = (cfg.now_s, 0)
    T, D = full.n_thr, full.D
    rows = _responsible(full)
    want = oracle_mod.Oracle(full).reconcile(now, rows=rows, nthreads=4, wide=True)
    # (what decides is the TOTAL of the requests a rank holds — 64 x 2^59 = 2^65 on rank 0 — not what one throttle matches)
    engines = [E.Engine.for_snapshot(shaped(cfg.shard(r, 2), r * 64)) for r in range(2)]
    try:
        for e in engines:
            e.set_exchange_world(2)
        # left to themselves: rank 0 refuses, rank 1 would have filled ONE block
        with pytest.raises(E.EngineError) as ei:
            engines[0].aggregate_launch()
        assert ei.value.code == -4 and "kt_set_wide_sums" in str(ei.value)
        engines[1].aggregate_launch()
        assert engines[1].pending_partial_words() == (T * (2 * D + 2), False)
        # agreed: two blocks on both ranks
        parts = []
        for e in engines:
            e.set_wide_sums(1)
            words = 2 * T * (2 * D + 2)
            part = torch.zeros(words, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            e.use_partial_buffer(part.data_ptr(), words)
            e.aggregate_launch()
            assert e.pending_partial_words() == (words, True)
            assert e.partial_used_buffer()[1] == words
            e.synchronize()
            parts.append(part)
        total = parts[0] + parts[1]
        torch.cuda.synchronize()
        for r, e in enumerate(engines):
            buf = total.clone()
            torch.cuda.synchronize()
            e.use_partial_buffer(buf.data_ptr(), buf.numel())
            e.finalize_launch(now, apply=False)
            got = e.reconcile_fetch()
            hi, _ = e.reconcile_fetch_used_hi()
            np.testing.assert_array_equal(got.used.v[rows], want.used.v[:len(rows)], err_msg=f"rank {r} used, low words")
            np.testing.assert_array_equal(hi[rows], want.used_hi[:len(rows)], err_msg=f"rank {r} used, high words")
            np.testing.assert_array_equal(got.used.count[rows], want.used.count[:len(rows)])
            np.testing.assert_array_equal(got.thrl_flag[rows], want.thrl_flag[:len(rows)])
            e.use_partial_buffer(None, 0)
    finally:
        for e in engines:
            e.close()
