#!/usr/bin/env python
"""bench.py — pod x throttle admission decisions/sec of the MI355X throttle-evaluation engine.

One "step" = one pass of the hot path over the whole synthetic snapshot resident in HBM:
    reconcile  (kt_aggregate -> [RCCL all-reduce of the per-throttle `used` partials when N>1] -> kt_finalize,
                result stored as the CR status)
  + check      (kt_prepare_check + kt_check: PreFilter for EVERY pod against EVERY throttle)
decisions per step = P_total x T  (the 5-state status of every (pod, throttle) pair is determined).

Weak scaling (default): every rank holds `pods_per_gpu` pod rows of a job with P_total = N x pods_per_gpu pods;
`--scaling strong` fixes P_total (the config's pod count; configs[4]: 10M) and gives every rank P_total / N rows.
Throttle tables are replicated; the only exchange is one int64 sum all-reduce of [T][2D+2] words — with N > 1 through
the engine's own RCCL communicator (kt_comm_*: what a Go host links), `--torch-comm` runs it through torch.distributed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    1: "configs[1]: 10k pods x 100 Throttles, D=4, single selectorTerm",
    2: "configs[2]: 1M pods x 1k Throttle+ClusterThrottle, D=8",
    3: "configs[3]: 1M pods x 1k throttles, temporaryThresholdOverrides active, D=8",
    4: "configs[4]: 10M pods x 10k throttles, multi-term OR-of-AND selectors, D=8 (pods row-sharded)",
}


def algorithmic_bytes(snap, n_pods, L):
    """SURVEY.md 8d / BASELINE.md: compulsory traffic of the logical records of one check pass."""
    D, T = snap.D, snap.n_thr
    b_pod = 8 + 4 * L + 8 * D
    b_out = 8
    n_req = len(snap.preq) + len(snap.nreq)
    thr_bytes = T * (16 + (8 * D + 12) * 3 + 4) + 16 * n_req
    return n_pods * (b_pod + b_out) + thr_bytes, b_pod, thr_bytes


def aggregate_bytes(snap, n_counted, L):
    D, T = snap.D, snap.n_thr
    n_req = len(snap.preq) + len(snap.nreq)
    return n_counted * (8 + 4 * L + 8 * D) + T * 16 + 16 * n_req + T * (2 * D + 1) * 8


# ---- self-verification of a run.  The finalize is REPLICATED: after the exchange every rank computes `used`,
# thresholds and throttled flags of every throttle from the same summed partials, so the fingerprints of those tables must be
# EQUAL over the ranks; the summary words belong to a rank's own pod rows and are fingerprinted per rank.  These three functions
# are what `bench.py --gpus N` runs after the timed region (and tests/test_bench_verify_cpu.py drives with fake ranks on CPU).
REPLICATED_FIELDS = ("used.v", "used.count", "used.present", "thrl_flag", "thrl_has", "thrl_pod", "error")


def result_hashes(rec, summary):
    """{"replicated": sha1 of the post-finalize per-throttle tables, "own": sha1 of this rank's summary words, "both": the
    fingerprint of the whole step as `results_sha1` has been since round 3}."""
    import hashlib
    import numpy as np
    h_rep, h_own, h_both = hashlib.sha1(), hashlib.sha1(), hashlib.sha1()
    for a in (rec.used.v, rec.used.count, rec.used.present, rec.thrl_flag, rec.thrl_has, rec.thrl_pod, rec.error):
        b = np.ascontiguousarray(a).tobytes()
        h_rep.update(b), h_both.update(b)
    b = np.ascontiguousarray(summary).tobytes()
    h_own.update(b), h_both.update(b)
    return {"replicated": h_rep.hexdigest()[:16], "own": h_own.hexdigest()[:16], "both": h_both.hexdigest()[:16]}


def ranks_agree(per_rank):
    """per_rank: the result_hashes() of every rank (+ "rank").  The replicated finalize must have left the same tables everywhere."""
    reps = {g["replicated"] for g in per_rank}
    return len(reps) == 1


def verify_against_oracle(full_snap, now, rec, throttle_rows, nthreads=None):
    """`rec`: a rank's post-finalize result (replicated).  The oracle reconciles the sampled throttles on the UNSHARDED snapshot —
    what the 8 ranks together must have computed; returns the list of mismatching (field, throttle row) pairs (empty = parity)."""
    import numpy as np
    from oracle import kt_oracle as O
    o = O.Oracle(full_snap)
    want = o.reconcile(now, rows=np.asarray(throttle_rows, dtype=np.int64), nthreads=nthreads or O.effective_cpus())
    bad = []
    for j, t in enumerate(throttle_rows):
        for name, g, w in (("used.v", rec.used.v[t], want.used.v[j]), ("used.count", rec.used.count[t], want.used.count[j]),
                           ("used.present", rec.used.present[t], want.used.present[j]), ("thrl_flag", rec.thrl_flag[t], want.thrl_flag[j]),
                           ("thrl_has", rec.thrl_has[t], want.thrl_has[j]), ("thrl_pod", rec.thrl_pod[t], want.thrl_pod[j]),
                           ("error", rec.error[t], want.error[j])):
            if not np.array_equal(np.asarray(g), np.asarray(w)):
                bad.append((name, int(t)))
    return bad


def extra_leg(index, min_seconds):
    """One more configuration timed on this GPU after the headline leg (N = 1 only): the same step — reconcile with APPLY +
    PreFilter sweep of every pod — for at least `min_seconds`, per-kernel HIP-event times from a second pass, the roofline
    fractions of both scans and the fingerprint of what the step left behind.  configs[4] is its first 1/8 shard (1.25M pod
    rows of the 10M-pod job's generator stream, all 10k throttles)."""
    import numpy as np
    import torch
    from kube_throttler_amd import engine as E, snapshot as S, workload as W
    cfg = W.preset(index)
    per_gpu = cfg.n_pods_total // 8 if index == 4 else cfg.n_pods_total
    cfg.n_pods_total = per_gpu
    cfg.pod_begin = 0
    cfg.n_pods = per_gpu
    t0 = time.time()
    snap = W.generate(cfg)
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED, device=torch.cuda.current_device())
    t_setup = time.time() - t0
    now = (cfg.now_s, 0)
    ts = torch.cuda.Stream()
    stream = ts.cuda_stream
    with torch.cuda.stream(ts):
        partial = torch.zeros(eng.partial_words(), dtype=torch.int64, device="cuda")
    ts.synchronize()
    eng.use_partial_buffer(partial.data_ptr(), partial.numel())

    def step():
        eng.reconcile_launch(now, True, stream)
        eng.check_launch(per_gpu, None, False, False, stream)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t_cal = time.perf_counter() - t0
    steps = 10 * int(max(1, min(10000, -(-(1.2 * min_seconds) // max(t_cal, 1e-6)))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    eng.timing_enable(False)
    k_ms = {}
    for name, fam in (("check", E.KERNEL_CHECK), ("aggregate", E.KERNEL_AGGREGATE), ("reduce", E.KERNEL_REDUCE),
                      ("finalize", E.KERNEL_FINALIZE), ("prepare", E.KERNEL_PREPARE)):
        tot, n = eng.timing_read(fam)
        k_ms[name] = tot / max(n, 1)
    rec_f = eng.reconcile_fetch()
    _, sm_f = eng.check_fetch(per_gpu, False)
    sha_both = result_hashes(rec_f, sm_f)["both"]
    flags = snap.pod_flags[:per_gpu]
    need = S.POD_VALID | S.POD_SCHED_MATCH | S.POD_SCHEDULED
    n_counted = int(((flags & need) == need).sum())
    chk_bytes, _, _ = algorithmic_bytes(snap, per_gpu, snap.L)
    agg_bytes = aggregate_bytes(snap, n_counted, snap.L)
    ms = elapsed * 1e3 / steps
    # the rocprofv3 averages of the same leg, when profiles/pmc_summary.json holds them for THIS library's sources
    prof_ms = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as fh:
            pmc = json.load(fh)
        key = "config%d_indexed" % index
        if pmc.get("_source", {}).get(key, {}).get("engine_version") == E.version():
            sym = lambda k: k.replace("_chunked", "").replace("_packed", "") if k.startswith("kt_aggregate") or k.startswith("kt_check") else k
            prof_ms = {}
            for name, fam in (("check", E.KERNEL_CHECK), ("aggregate", E.KERNEL_AGGREGATE), ("reduce", E.KERNEL_REDUCE), ("finalize", E.KERNEL_FINALIZE)):
                ns = pmc.get(key, {}).get(sym(eng.kernel_name(fam)), {}).get("rocprof_avg_ns")
                if ns and k_ms[name] > 0:
                    prof_ms[name] = round(ns * 1e-6, 6)
    except Exception:
        prof_ms = None
    frac = lambda nbytes, t_ms: round(nbytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if t_ms > 0 else 0.0
    out = {
        "workload": WORKLOADS[index] + (" — shard 0 of 8 on this GPU" if index == 4 else ""),
        "pods_per_gpu": per_gpu, "throttles": snap.n_thr, "steps": steps, "timed_region_s": round(elapsed, 6),
        "ms_per_step": round(ms, 6), "value": float(per_gpu) * float(snap.n_thr) * steps / elapsed, "unit": "decisions/s",
        "per_kernel_ms": {k: round(v, 6) for k, v in k_ms.items()},
        "kernels": {"check": eng.kernel_name(E.KERNEL_CHECK), "aggregate": eng.kernel_name(E.KERNEL_AGGREGATE),
                    "reduce": eng.kernel_name(E.KERNEL_REDUCE), "finalize": eng.kernel_name(E.KERNEL_FINALIZE)},
        "check_frac": frac(chk_bytes, k_ms["check"]), "aggregate_frac": frac(agg_bytes, k_ms["aggregate"]),
        "reconcile_frac": frac(agg_bytes, k_ms["aggregate"] + k_ms["reduce"] + k_ms["finalize"]),
        "step_frac": frac(chk_bytes + agg_bytes, ms),
        "algorithmic_bytes": {"check": chk_bytes, "aggregate": agg_bytes},
        "per_kernel_ms_rocprof": prof_ms or None,
        "results_sha1": sha_both, "setup_s": round(t_setup, 2),
    }
    eng.close()
    del partial
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs[i], i in 1..4 (default 2)")
    ap.add_argument("--pods-per-gpu", type=int, default=0, help="override the per-GPU pod rows")
    ap.add_argument("--variant", choices=["indexed", "dense"], default="indexed")
    ap.add_argument("--throttles", type=int, default=0,
                    help="measurements only: scale the config's throttle count (ClusterThrottles in proportion) — NOT a BASELINE config")
    ap.add_argument("--dims", type=int, default=0,
                    help="measurements only: resource dimensions per pod / throttle (up to 16) — NOT a BASELINE config")
    ap.add_argument("--labels", type=int, default=0,
                    help="measurements only: labels per pod (the label keys grow to twice that) — NOT a BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--verify", action="store_true", help="also bit-compare a pod sample with the oracle")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency leg (tools/latency_bench.py)")
    ap.add_argument("--native-comm", action="store_true",
                    help="exchange the partials with the engine's own RCCL communicator (kt_comm_*); the default when N > 1")
    ap.add_argument("--torch-comm", action="store_true", help="N > 1: run the exchange through torch.distributed instead")
    ap.add_argument("--overlap", action="store_true",
                    help="one GPU: the check of step i on a second stream beside the reconcile of step i + 1 (the PreFilter "
                         "and the controller run concurrently in the reference too); every check still reads the status its own "
                         "step's reconcile stored")
    ap.add_argument("--sweep", action="store_true",
                    help="one GPU: the step as ONE kt_sweep_launch — PreFilter sweep (against the stored status) and the reconcile scan "
                         "fused into one pass over the pod tables — instead of kt_reconcile_launch + kt_check_launch")
    ap.add_argument("--min-seconds", type=float, default=0.25,
                    help="the timed region runs at least this long: --steps is raised (to a multiple of itself) when K steps "
                         "would take less; the line reports the steps actually timed and the steps requested")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra legs (configs[3] and one configs[4] shard, timed after the headline leg on one GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: fixed rows per GPU (default); strong: the config's total pod count divided over the GPUs")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from kube_throttler_amd import engine as E, snapshot as S, workload as W

    # ---- workload: this rank's pod shard of a job with P_total = world x pods_per_gpu pods
    cfg = W.preset(args.config)
    if args.throttles:
        cfg.n_cluster = max(0, int(round(cfg.n_cluster * args.throttles / cfg.n_thr)))
        cfg.n_thr = args.throttles
    if args.dims:
        cfg.D = args.dims
    if args.labels:
        cfg.L = args.labels
        cfg.K = max(cfg.K, 2 * args.labels)
    if args.scaling == "strong":  # total work fixed: the config's pods (or --pods-per-gpu x 1 as the total) over N ranks
        total = args.pods_per_gpu or cfg.n_pods_total
        per_gpu = (total + world - 1) // world
    elif args.config == 4 and not args.pods_per_gpu:
        per_gpu = cfg.n_pods_total // 8  # the config is defined on 8 GPUs: 1.25M rows each
    else:
        per_gpu = args.pods_per_gpu or cfg.n_pods_total
    cfg.n_pods_total = per_gpu * world
    cfg.pod_begin = per_gpu * rank
    cfg.n_pods = per_gpu
    t0 = time.time()
    snap = W.generate(cfg)
    t_gen = time.time() - t0
    now = (cfg.now_s, 0)
    T, D, L = snap.n_thr, snap.D, snap.L
    P_total = cfg.n_pods_total

    variant = E.VARIANT_INDEXED if args.variant == "indexed" else E.VARIANT_DENSE
    t0 = time.time()
    eng = E.Engine.for_snapshot(snap, variant, device=local_rank)
    t_load = time.time() - t0
    # ONE explicit stream carries the whole step: the engine's kernels are enqueued on it through the C-ABI and the
    # RCCL all-reduce is issued with it as torch's current stream, so aggregate -> all-reduce -> finalize -> check are
    # ordered without any host synchronisation.  (The legacy null stream would NOT do: the engine substitutes its own
    # non-blocking stream for a null handle, which the collective would not be ordered against.)
    ts = torch.cuda.Stream()
    stream = ts.cuda_stream
    assert stream != 0

    with torch.cuda.stream(ts):
        partial = torch.zeros(eng.partial_words(), dtype=torch.int64, device="cuda")
    ts.synchronize()
    eng.use_partial_buffer(partial.data_ptr(), partial.numel())

    if world > 1 and not args.torch_comm:
        args.native_comm = True  # the product's own exchange is what a multi-GPU run measures
    if world > 1 and not args.native_comm:
        eng.set_exchange_world(world)  # the exact-range guard must cover the sum over all ranks
    native_comm_error = None
    if args.native_comm:  # the framework-free exchange: torch.distributed only carries the 128-byte id to the ranks
        try:
            ids = [E.Engine.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            eng.comm_init(rank, world, ids[0])
        except Exception as ex:  # the native exchange has never met more than one rank on hardware: say so and go on
            native_comm_error = "%s: %s" % (type(ex).__name__, ex)
        if world > 1:  # every rank takes the same exchange: one failure moves all of them to torch.distributed (also RCCL)
            errs = [None] * world
            dist.all_gather_object(errs, native_comm_error)
            native_comm_error = next((e for e in errs if e), None)
        if native_comm_error:
            if world == 1:
                raise SystemExit("kt_comm_init failed: " + native_comm_error)
            print("bench.py: native exchange unavailable (%s): using torch.distributed" % native_comm_error, file=sys.stderr)
            args.native_comm = False
            eng.set_exchange_world(world)

    xchg_events = []  # (start, stop) around the exchange, recorded only in the instrumented pass

    def step(timed_exchange=False):
        with torch.cuda.stream(ts):
            if args.native_comm or world > 1:
                eng.aggregate_launch(stream)
                if timed_exchange:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ts)
                if args.native_comm:
                    eng.comm_allreduce_partial(stream)  # ncclAllReduce(int64, sum) on the kernels' stream
                else:
                    dist.all_reduce(partial, op=dist.ReduceOp.SUM)  # RCCL over xGMI; int64 sums are order-independent
                if timed_exchange:
                    e1.record(ts)
                    xchg_events.append((e0, e1))
                eng.finalize_launch(now, True, stream)
            elif args.sweep:
                eng.sweep_launch(now, True, False, stream)  # check (stored status) + reconcile: one pass over the pod tables
                return
            else:
                eng.reconcile_launch(now, True, stream)  # one GPU: nothing to exchange between scan and finalize
            eng.check_launch(per_gpu, None, False, False, stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --overlap: two streams.  A carries the reconciles, B the checks; check(i) waits for finalize(i) (it reads the CheckRecs
    # that finalize left behind) and runs beside aggregate(i + 1), which reads nothing a check writes.  The engine keeps two
    # generations of CheckRecs once single-pod checks are in use (kt_engine.cpp finalize_locked: keep_prev), so finalize(i + 1)
    # writes the buffer check(i) is NOT reading; finalize(i + 2) rewrites the one check(i) read and therefore waits for it.
    overlap = args.overlap and world == 1 and not args.native_comm
    overlap_identical = None
    serial_step = step  # the instrumented pass below times the kernels one after the other in either mode
    if overlap:
        ts2 = torch.cuda.Stream()
        stream2 = ts2.cuda_stream
        ev_fin = [torch.cuda.Event(), torch.cuda.Event()]
        ev_chk = [torch.cuda.Event(), torch.cuda.Event()]
        step()  # the serial form once: its results are what the overlapped loop has to reproduce bit for bit
        fence()
        _, ref_summary = eng.check_fetch(per_gpu, False)
        ref_rec = eng.reconcile_fetch()
        eng.check_atomic(rows=np.zeros(1, np.int64), want_status=False)  # a PreFilter call: switches the double buffer on
        fence()
        n_over = [0]

        def step(timed_exchange=False):  # noqa: F811 — replaces the serial step from here on
            i = n_over[0]
            n_over[0] += 1
            if i >= 2:
                ts.wait_event(ev_chk[i % 2])  # check(i - 2) has released the CheckRec buffer finalize(i) rewrites
            with torch.cuda.stream(ts):
                eng.reconcile_launch(now, True, stream)
            ev_fin[i % 2].record(ts)
            ts2.wait_event(ev_fin[i % 2])
            with torch.cuda.stream(ts2):
                eng.check_launch(per_gpu, None, False, False, stream2)
            ev_chk[i % 2].record(ts2)

    for _ in range(args.warmup):
        step()
    fence()
    # A timed region of a millisecond (20 steps of 0.05 ms) rests on one noisy box: the requested K is raised to the multiple
    # of K that fills --min-seconds, from one untimed calibration pass of K steps (part of the warm-up).  Every rank
    # takes the same count (MAX over ranks).
    steps_requested = args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t_cal = time.perf_counter() - t0
    mult = 1
    if args.min_seconds > 0 and t_cal < args.min_seconds:
        mult = int(min(10000, -(-(1.2 * args.min_seconds) // max(t_cal, 1e-6))))  # (20 % margin: the calibration pass is short)
    if world > 1:
        tm = torch.tensor([mult], dtype=torch.int64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        mult = int(tm.item())
    args.steps = steps_requested * mult
    # the timed region carries no measurement hooks: the per-kernel HIP events are recorded in a second pass
    # (measured: recording them costs ~30 us per step; a hipGraph replay of the step is NOT faster than the five
    # back-to-back launches on this stack — 0.183 vs 0.178 ms at 1M x 1k — so the step is launched directly)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed * 1e3 / args.steps]
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_t = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(all_t, te)
        rank_ms = [float(t.item()) * 1e3 / args.steps for t in all_t]
        elapsed = max(float(t.item()) for t in all_t)  # MAX over ranks
    if overlap:
        _, got_summary = eng.check_fetch(per_gpu, False)
        got_rec = eng.reconcile_fetch()
        overlap_identical = bool(np.array_equal(got_summary, ref_summary) and all(
            np.array_equal(np.asarray(getattr(got_rec, f)), np.asarray(getattr(ref_rec, f)))
            for f in ("thrl_flag", "thrl_has", "thrl_pod", "error")) and np.array_equal(got_rec.used.v, ref_rec.used.v))
    # per-kernel durations: HIP events on the launch stream, same steps again
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(min(args.steps, 20)):
        serial_step(timed_exchange=True)
    fence()
    eng.timing_enable(False)
    exchange_ms = (sum(a.elapsed_time(b) for a, b in xchg_events) / len(xchg_events)) if xchg_events else None

    k_ms = {}
    for name, fam in (("check", E.KERNEL_CHECK), ("aggregate", E.KERNEL_AGGREGATE), ("reduce", E.KERNEL_REDUCE),
                      ("finalize", E.KERNEL_FINALIZE), ("prepare", E.KERNEL_PREPARE)):
        tot, n = eng.timing_read(fam)
        k_ms[name] = tot / max(n, 1)

    # a fingerprint of what the step left behind (every summary word, every throttle's used / flags): two builds that claim
    # the same results can be compared on full-size runs without an oracle pass (tools/gpu_ab.sh).  With several ranks the
    # replicated tables of every rank are compared (result_hashes / ranks_agree above): the first multi-GPU run verifies itself.
    rec_f = eng.reconcile_fetch()
    _, sm_f = eng.check_fetch(per_gpu, False)
    hashes = dict(result_hashes(rec_f, sm_f), rank=rank)
    results_sha1 = hashes["both"] if world == 1 else None
    per_rank_hashes = [hashes]
    if world > 1:
        per_rank_hashes = [None] * world
        dist.all_gather_object(per_rank_hashes, hashes)
    agree = ranks_agree(per_rank_hashes)
    if not agree:
        if rank == 0:
            print(json.dumps({"error": "the ranks' replicated finalize left DIFFERENT per-throttle tables: the exchange or a rank's scan is wrong",
                              "ranks_agree": False, "per_rank_hashes": per_rank_hashes}))
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)
    # --verify: a throttle sample of the (replicated) result against the oracle on the UNSHARDED snapshot — the partial sums of
    # all ranks together must be what one pass over every pod of the job gives
    oracle_verify = None
    if args.verify and rank == 0:
        vcfg = W.preset(args.config)
        for k in ("n_thr", "n_cluster", "D", "L", "K"):
            setattr(vcfg, k, getattr(cfg, k))
        vcfg.n_pods_total, vcfg.pod_begin, vcfg.n_pods = P_total, 0, P_total
        full = snap if world == 1 else W.generate(vcfg)
        need_t = S.THR_VALID | S.THR_RESPONSIBLE
        live = np.nonzero((full.thr_flags[:T] & need_t) == need_t)[0]
        sample_t = live[np.linspace(0, len(live) - 1, min(len(live), 48)).astype(np.int64)]
        t0v = time.time()
        bad = verify_against_oracle(full, now, rec_f, sample_t)
        oracle_verify = {"throttles": int(len(sample_t)), "pods": int(P_total), "mismatches": bad[:8], "ok": not bad, "seconds": round(time.time() - t0v, 1)}
    if world > 1:
        okv = [oracle_verify]
        dist.broadcast_object_list(okv, src=0)
        oracle_verify = okv[0]
    if oracle_verify is not None and not oracle_verify["ok"]:
        if rank == 0:
            print(json.dumps({"error": "the reconciled status differs from the oracle on the unsharded snapshot", "oracle_verify": oracle_verify}))
        if world > 1:
            dist.destroy_process_group()
        sys.exit(4)

    # what every rank's exchange and scans looked like — and the ranks must agree on the words of the partial buffer (the
    # all-reduce sums them position by position: a rank with another throttle set or another sum form would corrupt every
    # other rank's `used` silently)
    index_stats = eng.index_stats()
    per_rank_index = None
    try:
        eng.aggregate_launch(stream)
        torch.cuda.synchronize()
        pending = eng.pending_partial_words()
    except Exception as ex:
        pending = (-1, repr(ex))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "partial_words": pending[0], "wide": pending[1], **index_stats})
        per_rank_index = gathered
        if len({(g["partial_words"], g["wide"]) for g in gathered}) != 1:
            if rank == 0:
                print(json.dumps({"error": "the ranks disagree on kt_partial_words: the all-reduce would mix different buffers",
                                  "per_rank": gathered}))
            dist.destroy_process_group()
            sys.exit(2)

    per_rank_kernel_ms = None
    if world > 1:  # every rank's kernel times, so that the first multi-GPU run explains itself
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: round(v, 6) for k, v in k_ms.items()} | {"exchange": None if exchange_ms is None else round(exchange_ms, 6)})
        per_rank_kernel_ms = gathered

    decisions_per_step = float(P_total) * float(T)
    ms_per_step = elapsed * 1e3 / args.steps
    value = decisions_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel (per launch, this rank's shard)
    flags = snap.pod_flags[:per_gpu]
    need = S.POD_VALID | S.POD_SCHED_MATCH | S.POD_SCHEDULED
    n_counted = int(((flags & need) == need).sum())
    chk_bytes, b_pod, thr_bytes = algorithmic_bytes(snap, per_gpu, L)
    agg_bytes = aggregate_bytes(snap, n_counted, L)
    # the dominant kernel = the one that carries most of the step's ALGORITHMIC bytes (what a roofline is about).  The two
    # scans take the same time within run-to-run noise at 1M x 1k (31-32 us each), so "the slower one" would flip from
    # run to run; SURVEY.md 8d makes the check the primary rate and the aggregation the secondary one.  Both kernels are
    # reported in full below, and `step` prices the whole step against the roofline.
    dominant = "check" if chk_bytes >= agg_bytes else "aggregate"
    dom_bytes = chk_bytes if dominant == "check" else agg_bytes
    fused_sweep = eng.kernel_name(E.KERNEL_CHECK) == "kt_sweep_bitmap"
    if fused_sweep:  # ONE launch is both scans: it processes the check's AND the aggregation's units (SURVEY.md 8d figures of both)
        dominant, dom_bytes = "check", chk_bytes + agg_bytes
    achieved = dom_bytes / (k_ms[dominant] * 1e-3) / 1e9 if k_ms[dominant] > 0 else 0.0
    # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE per launch, tools/profile.sh + tools/pmc_summary.py) is only quoted when it
    # was measured on the SAME kernel sources as the library that just ran (kt_version() carries their hash)
    traffic = None
    engine_version = E.version()
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    pmc_kernels = {}
    pmc_source = None
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            key = f"config{args.config}_{args.variant}"
            if args.pods_per_gpu:  # e.g. config2_indexed_4M: the point past the 256 MiB Infinity Cache
                key += "_%dM" % (args.pods_per_gpu // 1000000) if args.pods_per_gpu % 1000000 == 0 else "_%d" % args.pods_per_gpu
            if args.dims or args.labels or args.throttles:  # not a profiled configuration: no PMC figures are quoted
                key += "_custom"
            src = pmc.get("_source", {}).get(key)
            if isinstance(src, dict) and src.get("engine_version") == engine_version:
                pmc_kernels = pmc.get(key, {})
                pmc_source = "profiles/%s_* (same kernel sources: %s)" % (src.get("tag"), engine_version.split("src=")[-1])
        except Exception:
            pmc_kernels = {}
    dom_kernel = eng.kernel_name(E.KERNEL_CHECK if dominant == "check" else E.KERNEL_AGGREGATE)
    # "..._chunked" / "..._packed" are the same kernel symbol (several index chunks / the packed fold instantiation)
    sym = lambda k: k.replace("_chunked", "").replace("_packed", "") if k.startswith("kt_aggregate") or k.startswith("kt_check") else k
    if sym(dom_kernel) in pmc_kernels:
        traffic = pmc_kernels[sym(dom_kernel)].get("hbm_bytes_per_launch")

    SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; MI355X peak engine clock (MI355X_MICROARCH.md)

    def bound_of(kn, ms_events):
        """What bounds the kernel, from the hash-gated profile of the SAME sources (profiles/pmc_summary.json): actual HBM
        traffic against the roofline, VALU issue time (4 cycles per wave-level instruction assumed — an upper bound: the logic /
        add / bitop3 instructions issue in 2.4-2.7, the three-operand and multiplier ones in 4.1-4.5, profiles/r04_valu_rates.txt) and LDS
        bank-conflict cycles per LDS-active cycle.  Durations: the rocprofv3 average when the profile has one."""
        rec = pmc_kernels.get(sym(kn))
        if not rec:
            return None
        ns = rec.get("rocprof_avg_ns")
        dur = ns * 1e-9 if ns else ms_events * 1e-3
        out = {"profile": pmc_source, "rocprof_avg_ms": None if not ns else round(ns * 1e-6, 6)}
        if rec.get("hbm_bytes_per_launch") and dur > 0:
            out["hbm_actual_bytes"] = rec["hbm_bytes_per_launch"]
            out["hbm_actual_frac"] = round(rec["hbm_bytes_per_launch"] / dur / 1e9 / HBM_PEAK_GBS, 4)
        sq = rec.get("sq") or {}
        if sq.get("SQ_INSTS_VALU") and dur > 0:
            out["valu_busy"] = round(sq["SQ_INSTS_VALU"] * 4.0 / (SIMDS * dur * CLOCK_HZ), 4)
            out["valu_insts_per_launch"] = sq["SQ_INSTS_VALU"]
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_conflict_frac"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"], 4)
        if sq.get("SQ_WAVE_CYCLES"):
            out["wave_wait_frac"] = round(sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"], 4)
        return out

    def kernel_roofline(name, fam, nbytes):
        ms = k_ms[name]
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        kn = eng.kernel_name(fam)
        d = {"kernel": kn, "avg_launch_ms": round(ms, 6), "algorithmic_bytes_per_launch": nbytes,
             "achieved": round(gbs, 3), "frac": round(gbs / HBM_PEAK_GBS, 6),
             "traffic_profiled": pmc_kernels.get(sym(kn), {}).get("hbm_bytes_per_launch"), "bound_by": bound_of(kn, ms)}
        ns = pmc_kernels.get(sym(kn), {}).get("rocprof_avg_ns")
        if ns:  # the un-instrumented duration (same sources): the HIP-event figure above carries the events' own cost
            d["frac_rocprof"] = round(nbytes / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 6)
        return d

    # reconcile = the three launches that turn pods into stored status (aggregate scan + slab reduction + finalize; the
    # fused kernel reports all of it under `aggregate`) against the aggregation's algorithmic bytes
    rec_ms = k_ms["aggregate"] + k_ms["reduce"] + k_ms["finalize"]
    slowest = max(("check", "aggregate", "reduce", "finalize"), key=lambda k: k_ms[k])
    fam_of = {"check": E.KERNEL_CHECK, "aggregate": E.KERNEL_AGGREGATE, "reduce": E.KERNEL_REDUCE, "finalize": E.KERNEL_FINALIZE}
    # HIP events around every launch keep consecutive kernels from overlapping their ramps: the instrumented pass is a few
    # microseconds longer per step than the timed region, so these per-kernel figures are upper bounds (their sum may
    # exceed ms_per_step); the rocprofv3 averages of the same sources, when profiled, are quoted beside them
    prof_ms = {}
    for name, fam in (("check", E.KERNEL_CHECK), ("aggregate", E.KERNEL_AGGREGATE), ("reduce", E.KERNEL_REDUCE), ("finalize", E.KERNEL_FINALIZE)):
        ns = pmc_kernels.get(sym(eng.kernel_name(fam)), {}).get("rocprof_avg_ns")
        if ns and k_ms[name] > 0:
            prof_ms[name] = round(ns * 1e-6, 6)
    roofline = {
        "bound": "hbm", "kernel": dom_kernel,
        "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": traffic, "traffic_source": pmc_source, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(k_ms[dominant], 6),
        "per_kernel_ms": {k: round(v, 6) for k, v in k_ms.items()},
        "per_kernel_ms_note": "HIP events on the launch stream, instrumented pass (upper bounds: the events keep kernels from overlapping)",
        "per_kernel_ms_rocprof": prof_ms or None,
        "launch_gaps_ms": round(ms_per_step - sum(prof_ms.values()), 6) if len(prof_ms) >= 2 else None,
        "dominant_by": "algorithmic bytes per launch",
        "fused_sweep": fused_sweep,
        "check": kernel_roofline("check", E.KERNEL_CHECK, chk_bytes + agg_bytes if fused_sweep else chk_bytes),
        "aggregate": kernel_roofline("aggregate", E.KERNEL_AGGREGATE, agg_bytes),
        "reconcile": {"kernels": [eng.kernel_name(E.KERNEL_AGGREGATE), eng.kernel_name(E.KERNEL_REDUCE), eng.kernel_name(E.KERNEL_FINALIZE)],
                      "ms": round(rec_ms, 6), "algorithmic_bytes": agg_bytes,
                      "frac": round(agg_bytes / (rec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if rec_ms > 0 else 0.0},
        # the other reading of "dominant": the launch that takes longest (it flips between the two scans from run to run)
        "slowest": {"kernel": eng.kernel_name(fam_of[slowest]), "family": slowest, "avg_launch_ms": round(k_ms[slowest], 6),
                    "frac": round({"check": chk_bytes, "aggregate": agg_bytes}.get(slowest, 0) / (k_ms[slowest] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
                    if k_ms[slowest] > 0 else 0.0},
        # the whole step (both scans + slab reduction + finalize, launch gaps included) against the same roofline
        "step": {"algorithmic_bytes": chk_bytes + agg_bytes, "ms": round(ms_per_step, 6),
                 "frac": round((chk_bytes + agg_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if ms_per_step > 0 else 0.0},
    }

    # ---- CPU baseline: the C restatement of the reference algorithm on this box's host cores (rank 0, N=1)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import kt_oracle as O  # test infrastructure: used here ONLY as the timed CPU baseline
        rec = eng.reconcile_fetch()
        snap.apply_status(rec.used, rec.calc, rec.calc_updated, rec.thrl_flag, rec.thrl_has, rec.thrl_pod, rec.error)
        o = O.Oracle(snap)
        cores = O.effective_cpus()  # the affinity mask capped by the cgroup CPU quota: the threads that really run side by side
        # bounded sample: grow it until one pass costs a few seconds of wall time, then repeat to ~cpu_seconds
        n_sample = min(per_gpu, 16384)
        while True:
            sample = np.linspace(0, per_gpu - 1, n_sample).astype(np.int64)
            t0 = time.perf_counter()
            _, sm_cpu = o.check(rows=sample, want_status=False, nthreads=cores, mimic_log_args=True)
            dt = time.perf_counter() - t0
            if dt >= min(3.0, args.cpu_seconds) or n_sample >= per_gpu:
                break
            n_sample = int(min(per_gpu, max(n_sample * 2, n_sample * 3.0 / max(dt, 1e-3))))
        reps, total = 1, dt
        while total < args.cpu_seconds and reps < 64:
            t0 = time.perf_counter()
            o.check(rows=sample, want_status=False, nthreads=cores, mimic_log_args=True)
            total += time.perf_counter() - t0
            reps += 1
        cpu_baseline = {
            "value": round(len(sample) * T * reps / total, 1), "unit": "decisions/s", "cores": cores, "kind": "port",
            "sample": f"PreFilter (CheckThrottled x2 + CheckThrottledFor) for {len(sample)} of {per_gpu} pods x {T} "
                      f"throttles, {reps} passes, {total:.1f}s total, C restatement of the reference algorithm "
                      f"(oracle/kt_oracle.c, OpenMP over pods on {cores} threads, klog eager-argument work included); "
                      f"NOT the reference Go binary",
        }
        # the single-pod yardstick for the latency leg: PreFilter of ONE pod on ONE core, reference loop shape
        one = sample[:256]
        t0 = time.perf_counter()
        o.check(rows=one, want_status=False, nthreads=1, mimic_log_args=True)
        cpu_baseline["prefilter_one_pod_us_1core"] = round((time.perf_counter() - t0) / len(one) * 1e6, 1)
        if args.verify:
            _, sm_gpu = eng.check(rows=sample, want_status=False)
            assert np.array_equal(sm_gpu, sm_cpu), "GPU summaries differ from the oracle on the sample"

    # ---- what the scheduler would feel: single-call latencies through the C-ABI (rank 0, N=1; after the timed region)
    latency = None
    if rank == 0 and world == 1 and not args.no_latency and not args.native_comm:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import latency_bench
            eng.use_partial_buffer(None, 0)
            latency = latency_bench.measure(eng, snap, n_check=4000, n_upsert=200, now=now)
        except Exception as ex:  # never lose the bench line over the side measurement
            latency = {"error": repr(ex)}
        if isinstance(latency, dict) and "sweep" in latency:
            # the event-driven step: in the reference EVERY reconcile follows an event (throttle_controller.go:400-536);
            # a sweep right after ONE pod upsert, host wall clock incl. launches, against the same roofline
            sw = latency["sweep"]
            roofline["step_after_event"] = {
                "ms": sw["after_pod_event_ms"], "steady_ms": sw["steady_ms"],
                "ratio": round(sw["after_pod_event_ms"] / sw["steady_ms"], 3) if sw["steady_ms"] > 0 else None,
                "frac": round((chk_bytes + agg_bytes) / (sw["after_pod_event_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
                if sw["after_pod_event_ms"] > 0 else 0.0}

    if rank == 0:
        out = {
            "metric": "pod_throttle_decisions_per_sec", "value": value, "unit": "decisions/s", "n_gpus": world,
            "steps": args.steps, "steps_requested": steps_requested, "timed_region_s": round(elapsed, 6),
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config] + (" — throttle count overridden (--throttles): not a BASELINE config" if args.throttles else "")
                                   + (" — dimensions / labels overridden (--dims / --labels): not a BASELINE config" if args.dims or args.labels else ""),
                       "pods_total": P_total, "pods_per_gpu": per_gpu,
                       "throttles": T, "dims": D, "labels_per_pod": L, "namespaces": snap.n_ns,
                       "step": "sweep(check all pods against the stored status + aggregate, one pass)+finalize(apply)" if args.sweep
                               else "reconcile(aggregate+allreduce+finalize,apply)+check(all pods)",
                       "streams": "check(i) on a second stream beside reconcile(i+1); check(i) after finalize(i)" if overlap else "one",
                       "overlap_identical_to_serial": overlap_identical, "results_sha1": results_sha1,
                       "kernel_variant": args.variant, "parallelism": f"pods row-sharded x{world}, throttles replicated",
                       "exchange": "kt_comm (RCCL, native)" if args.native_comm else ("torch.distributed (RCCL)" if world > 1 else "none"),
                       "native_exchange_error": native_comm_error,
                       "generate_s": round(t_gen, 2), "load_s": round(t_load, 2), "engine_version": engine_version},
            "per_rank_ms_per_step": [round(x, 6) for x in rank_ms],
            "exchange_ms": None if exchange_ms is None else round(exchange_ms, 6),
            "per_rank_kernel_ms": per_rank_kernel_ms,
            "ranks_agree": agree, "per_rank_hashes": per_rank_hashes, "oracle_verify": oracle_verify,
            "index": index_stats, "partial_words": pending[0], "per_rank_index": per_rank_index,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "latency": latency,
        }
    if args.native_comm:
        eng.comm_destroy()
    eng.close()
    if rank == 0:
        # the other single-GPU BASELINE configurations, driver-observed instead of builder-claimed: after the headline leg
        # (its engine is closed), on this GPU, each for at least --min-seconds
        if world == 1 and not args.no_extra and args.config == 2 and not args.pods_per_gpu and not args.throttles and not args.dims and not args.labels and args.variant == "indexed":
            extra = {}
            for idx, key in ((3, "configs[3]"), (4, "configs[4] shard")):
                try:
                    extra[key] = extra_leg(idx, args.min_seconds)
                except Exception as ex:  # never lose the headline line over an extra leg
                    extra[key] = {"error": repr(ex)}
            out["extra_configs"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
