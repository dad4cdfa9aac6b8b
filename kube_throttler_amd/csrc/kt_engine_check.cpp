// kt_engine_check.cpp — KubeThrottler.PreFilter (plugin.go:148-215) behind the C-ABI: the sweep over every pod, checks of a few
// pods (kt_check: the scheduler's call), admission queues with reservation, affected pods, and the paged forms for more than
// 16 resource names.
#include "kt_engine_impl.h"

// ---------------------------------------------------------------------------------------------------
// check
// ---------------------------------------------------------------------------------------------------
// allow_small: false for callers that go on working on the device-side rows / summaries (kt_admit_launch)
// The CheckRecs only depend on (stored status, reserved amounts, isThrottledOnEqual): rebuilt when one of them changed
// since they were last built (by kt_prepare_check or by kt_finalize with APPLY)
static int32_t ensure_check_recs(kt_engine* e, int32_t on_equal, int DT, hipStream_t s) {
  if (e->recs_valid && e->recs_eq == (on_equal != 0) && e->recs_DT == DT) return KT_OK;
  recs_invalidate_and_drain(e);  // rebuilt in place
  {
    TimedLaunch tl(e, KT_KERNEL_PREPARE, s);
    kt::launch_prepare_check(e->tt, e->thr_rows_hi, e->D, DT, on_equal != 0, e->recs_ptr(), req_bound(e), s);
  }
  if (e->few_ready) KT_HIP(e, hipEventRecord(e->recs_ev[e->recs_cur], s));
  std::lock_guard<std::mutex> g(e->recs_mu);
  e->recs_ev_pending[e->recs_cur] = e->few_ready;
  e->recs_prev_valid = false;  // records of an older status / other on_equal: not a substitute any more
  ++e->recs_seq[e->recs_cur];
  e->recs_valid = true;
  e->recs_eq = on_equal != 0;
  e->recs_DT = DT;
  return KT_OK;
}

static int32_t check_launch_locked(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint32_t flags,
                                   hipStream_t s, bool allow_small = true) {
  if (pod_rows) {
    for (int64_t i = 0; i < n; ++i)
      if (pod_rows[i] < 0 || pod_rows[i] >= e->cfg.pod_capacity)
        return e->fail(KT_ERR_OUT_OF_RANGE, "pod row %lld", (long long)pod_rows[i]);
  } else if (n > e->cfg.pod_capacity) {
    return e->fail(KT_ERR_OUT_OF_RANGE, "n=%lld > pod_capacity", (long long)n);
  }
  int32_t rc = ensure_ready(e, s);
  if (rc != KT_OK) return rc;
  const bool want_status = (flags & KT_CHECK_STATUS_MATRIX) != 0;
  const size_t T = (size_t)e->thr_rows_hi;
  if (e->d_summary.cap < (size_t)n + 1 || (want_status && e->d_status.cap < (size_t)n * T + 64) ||
      (pod_rows && e->d_rows.cap < (size_t)n + 1)) {
    if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));  // buffers may still be in use
    KT_HIP(e, e->d_summary.reserve((size_t)n + 1));
    if (want_status) KT_HIP(e, e->d_status.reserve((size_t)n * T + 64));  // slack: kt_admit_sequential reads rows 16 bytes at a time
    if (pod_rows) KT_HIP(e, e->d_rows.reserve((size_t)n + 1));
  }
  // a handful of pods (one PreFilter call): rows by value, one workgroup per index chunk, summaries to pinned memory
  const bool small = allow_small && e->cfg.kernel_variant != 1 && n > 0 && n <= kt::kCheckSmallMax;
  kt::CheckSmall sm{};
  if (small) {
    if (!e->h_small) {
      KT_HIP(e, hipHostMalloc((void**)&e->h_small, (size_t)kt::kCheckSmallMax * 8, hipHostMallocMapped));
      KT_HIP(e, e->d_ticket.reserve(16));
      KT_HIP(e, hipMemsetAsync(e->d_ticket.p, 0, 16 * 4, s));
    }
    sm.ticket = e->d_ticket.p;
    sm.host_summary = e->h_small;
    if (pod_rows && n <= 8) {
      sm.n_inline = (uint32_t)n;
      for (int64_t k = 0; k < 8; ++k) sm.inline_rows[k] = pod_rows[k < n ? k : n - 1];
    }
  }
  if (pod_rows && n && !sm.n_inline) {
    KT_HIP(e, hipMemcpyAsync(e->d_rows.p, pod_rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
    KT_HIP(e, hipStreamSynchronize(s));  // caller memory must not be referenced after return
  }
  // the record layout follows the scan kernel that will read it
  const int DT = e->cfg.kernel_variant == 1 ? kt::dt_bucket(e->D) : kt::dt_bucket_ix(e->D);
  if ((rc = ensure_check_recs(e, on_equal, DT, s)) != KT_OK) return rc;
  {
    TimedLaunch tl(e, KT_KERNEL_CHECK, s);
    if (e->cfg.kernel_variant == 1)
      kt::launch_check_dense(e->pods, n, pod_rows ? e->d_rows.p : nullptr, e->sp, e->uses_keys, e->recs_ptr(),
                             e->d_summary.p, want_status ? e->d_status.p : nullptr, s),
          e->last_kernel[KT_KERNEL_CHECK] = "kt_check_dense";
    else {
      // a sweep over every row of a multi-chunk index runs in namespace order (results stay indexed by pod row)
      const bool by_ns = !pod_rows && !small && n == e->pod_rows_hi && (e->dindex.n_chunks > 1 || e->sw[kSw_FORCE_NS_ORDER]) && !e->sw[kSw_NO_NS_ORDER];
      if (by_ns && (rc = settle_view_patches(e, s)) != KT_OK) return rc;
      if (by_ns && (!e->order_all_valid || e->view_rows_a != e->pod_rows_hi)) {
        KT_HIP(e, e->d_order_all.reserve((size_t)e->pod_rows_hi + 1));
        KT_HIP(e, e->d_ns_cursor.reserve((size_t)e->sp.n_ns + 1));
        KT_HIP(e, e->d_n_all.reserve(1));
        kt::launch_order_rows_by_ns(e->pods, e->pod_rows_hi, /*countable_only=*/false, (uint32_t)e->sp.n_ns,
                                    e->d_ns_cursor.p, e->d_order_all.p, e->d_n_all.p, s);
        KT_HIP(e, hipGetLastError());
        e->range_a_G = 0;
        if (!e->sw[kSw_NO_WG_RANGES]) {  // every row is listed: the list holds pod_rows_hi records
          // (planned on the host from a copy of the namespace ends — a view build is not a per-step cost, and the one GPU
          //  thread the plan used to run on took 388 us, longer than the synchronisation and the walk here)
          e->range_a_G = kt::check_sweep_blocks(e->pod_rows_hi);
          KT_HIP(e, e->d_range_a.reserve((size_t)e->range_a_G + 2));
          e->h_ns_end.resize((size_t)e->sp.n_ns + 1);
          KT_HIP(e, hipMemcpyAsync(e->h_ns_end.data(), e->d_ns_cursor.p, (size_t)e->sp.n_ns * 8, hipMemcpyDeviceToHost, s));
          KT_HIP(e, hipStreamSynchronize(s));
          e->h_range.resize((size_t)e->range_a_G + 2);
          kt::plan_wg_ranges(e->h_ns_end.data(), (uint32_t)e->sp.n_ns, e->pod_rows_hi, e->range_a_G, e->h_range.data());
          KT_HIP(e, hipMemcpyAsync(e->d_range_a.p, e->h_range.data(), e->h_range.size() * 4, hipMemcpyHostToDevice, s));
          KT_HIP(e, hipStreamSynchronize(s));
        }
        const size_t na = (size_t)e->pod_rows_hi + 1;
        KT_HIP(e, e->d_va_meta.reserve(na));
        KT_HIP(e, e->d_va_latom.reserve(na * (size_t)e->pods.LA));
        KT_HIP(e, e->d_carry.reserve(na));
        KT_HIP(e, e->d_pos_a.reserve(na));
        KT_HIP(e, e->d_view_dirty.reserve(4));
        if (!e->view_check_dirty) KT_HIP(e, hipMemsetAsync(e->d_view_dirty.p, 0, 4, s));
        KT_HIP(e, hipMemsetAsync(e->d_pos_a.p, 0xFF, na * 4, s));
        kt::launch_build_scan_view(e->pods, e->pod_rows_hi, e->d_order_all.p, e->d_va_meta.p, e->d_va_latom.p, nullptr, s, nullptr, nullptr, e->d_pos_a.p);
        KT_HIP(e, hipGetLastError());
        e->view_rows_a = e->pod_rows_hi;
        e->order_all_valid = true;
      }
      kt::CheckByNs view{e->d_va_meta.p, e->d_va_latom.p, e->d_carry.p};
      if (by_ns && e->range_a_G) view.wg_range = e->d_range_a.p, view.wg_range_G = e->range_a_G;
      if (by_ns && !want_status && e->dindex.n_slow == 0 && e->n_overflow == 0 && e->dindex.n_chunks > 1 && !e->sw[kSw_NO_VERDICT_IMAGES]) {
        // the lean sweep of a multi-chunk program: TermInfo + WordVerdict of every word once per generation of CheckRecs
        // (one small launch) instead of once per (workgroup, chunk) — 256 x ~15 rebuilds of the same words
        const int b = e->recs_cur;
        if (e->wvimg_seq[b] != e->recs_seq[b] || e->wvimg_gen[b] != e->program_gen || e->wvimg_DT[b] != DT) {
          KT_HIP(e, e->d_wvimg[b].reserve(kt::verdict_images_bytes(e->dindex.bm_words, e->D)));
          TimedLaunch tl2(e, KT_KERNEL_PREPARE, s);
          kt::launch_build_verdict_images(e->dindex, e->dindex.bm_words, e->recs_ptr(), e->thr_rows_hi, e->D, e->d_wvimg[b].p, s);
          e->wvimg_seq[b] = e->recs_seq[b], e->wvimg_gen[b] = e->program_gen, e->wvimg_DT[b] = DT;
        }
        view.wv_img = e->d_wvimg[b].p, view.wv_total_words = e->dindex.bm_words;
      }
      const char* k = kt::launch_check_indexed(e->pods, n, by_ns ? e->d_order_all.p : pod_rows ? e->d_rows.p : nullptr, e->sp, e->d_sp.p, e->dindex,
                                               e->recs_ptr(), e->d_summary.p, want_status ? e->d_status.p : nullptr, s,
                                               small ? &sm : nullptr, e->n_overflow != 0, by_ns ? &view : nullptr, e->sw[kSw_CHECK_ONE_PER_CU]);
      if (!k) return e->fail(KT_ERR_UNSUPPORTED, "%d throttle rows exceed the indexed check kernel's LDS budget (use kernel_variant 1)", e->thr_rows_hi);
      e->last_kernel[KT_KERNEL_CHECK] = k;
    }
  }
  KT_HIP(e, hipGetLastError());
  e->check_n = n;
  e->check_in_h_small = small;
  e->check_T = e->thr_rows_hi;
  e->check_has_status = want_status;
  e->check_ready = true;
  e->last_stream = s;
  return KT_OK;
}

int32_t kt_check_launch(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint32_t flags, void* stream) {
  if (!e || n < 0) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  return check_launch_locked(e, n, pod_rows, on_equal, flags, pick_stream(e, stream));
}

// kt_sweep_launch — the PreFilter sweep of every pod row against the STORED status and the reconcile of every throttle
// as one pass over the pod tables (kt_check_bitmap's AGG instantiation: one chunk prologue and one selector scan per pod
// where kt_check_launch + kt_reconcile_launch make two), then kt_reduce_finalize_packed.  Results are read with
// kt_check_fetch / kt_reconcile_fetch and are bit for bit those of kt_check_launch(all rows) followed by
// kt_reconcile_launch — which is also what runs when the fused kernel does not apply (several index chunks, a slow list,
// pods whose atoms overflow their row, requests that do not pack, wide sums, an incremental engine, the dense variant).
int32_t kt_sweep_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t on_equal, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  hipStream_t s = pick_stream(e, stream);
  int32_t rc = ensure_ready(e, s);
  if (rc != KT_OK) return rc;
  const int64_t n = e->pod_rows_hi;
  auto one_after_the_other = [&]() -> int32_t {
    int32_t r = check_launch_locked(e, n, nullptr, on_equal, 0u, s);
    if (r != KT_OK) return r;
    if ((r = aggregate_locked(e, s, /*allow_fused=*/true)) != KT_OK) return r;
    return finalize_locked(e, now_s, now_ns, flags, s, /*consume=*/!e->incremental);
  };
  bool fused = e->cfg.kernel_variant != 1 && !e->incremental && e->dindex.n_chunks == 1 && e->dindex.n_slow == 0 && !e->hindex.has_slow && !e->dindex.has_long &&
               e->n_overflow == 0 && e->thr_rows_hi > 0 && n > 0 && kt::dt_bucket_ix(e->D) == 8 && !e->sw[kSw_NO_SWEEP] &&
               !e->sw[kSw_NO_FUSED] && !e->sw[kSw_NO_PACK];
  if (!fused) return one_after_the_other();
  if ((rc = request_sums_in_range(e, s)) != KT_OK) return rc;
  if (e->wide) return one_after_the_other();
  // the packed fold's plan for THIS scan: every row of [0, n) in row order, aggregate_blocks(n) workgroups
  const int nb = kt::aggregate_blocks(n);
  kt::PackPlan plan = kt::make_pack_plan(e->D, e->max_abs, e->or_abs, e->neg_seen, kt::aggregate_slab_pods(n, nb), /*pad_odd=*/true, /*max_words=*/4);
  if (!plan.nw || plan.rec_bytes > kt::agg_rec_bytes(e->D, false)) return one_after_the_other();  // (slab areas hold plain records)
  const size_t words = (size_t)e->thr_rows_hi * kt::partial_stride(e->D);
  if (e->ext_partial && (int64_t)words > e->ext_partial_words)
    return e->fail(KT_ERR_OUT_OF_RANGE, "caller partial buffer holds %lld words, %lld needed", (long long)e->ext_partial_words, (long long)words);
  if (e->d_summary.cap < (size_t)n + 1) {
    if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));  // the buffer may still be in use
    KT_HIP(e, e->d_summary.reserve((size_t)n + 1));
  }
  const int DT = kt::dt_bucket_ix(e->D);
  if ((rc = ensure_check_recs(e, on_equal, DT, s)) != KT_OK) return rc;
  e->fused_pending = false;
  e->agg_wide = false;
  if (words && e->clean_partial != (const void*)e->partial()) KT_HIP(e, hipMemsetAsync(e->partial(), 0, words * 8, s));
  e->clean_partial = nullptr;
  kt::AggScan sc;
  if ((rc = slab_tags(e, sc, s)) != KT_OK) return rc;
  int launched = 0;
  const char* k;
  {
    TimedLaunch tl(e, KT_KERNEL_CHECK, s);
    k = kt::launch_sweep_indexed(e->pods, n, e->sp, e->d_sp.p, e->dindex, e->recs_ptr(), e->d_summary.p, plan, e->d_slab.p, sc.slab_tag, sc.epoch,
                                 &launched, s);
  }
  if (!k) return one_after_the_other();  // (LDS: check tables + fold tables of this program do not fit one workgroup)
  if (launched != nb) return e->fail(KT_ERR_DEVICE, "kt_sweep_launch: %d workgroups launched, the packed fields were planned for %d", launched, nb);
  KT_HIP(e, hipGetLastError());
  e->last_kernel[KT_KERNEL_CHECK] = k;
  e->last_kernel[KT_KERNEL_AGGREGATE] = "(in kt_sweep_bitmap)";
  e->last_kernel[KT_KERNEL_REDUCE] = "(in kt_reduce_finalize_packed)";
  e->check_n = n, e->check_in_h_small = false, e->check_T = e->thr_rows_hi, e->check_has_status = false, e->check_ready = true;
  e->fused_pending = true, e->fused_nb = launched, e->fused_epoch = sc.epoch, e->fused_pack = plan;
  e->agg_pending = true, e->agg_words = words, e->agg_gen = e->program_gen;
  e->last_stream = s;
  return finalize_locked(e, now_s, now_ns, flags, s, /*consume=*/true);
}

// ---------------------------------------------------------------------------------------------------
// sequential admission with reservation
// ---------------------------------------------------------------------------------------------------
int32_t kt_admit_launch(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint32_t flags, void* stream) {
  if (!e || n < 0) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  hipStream_t s = pick_stream(e, stream);
  if (e->wide)
    return e->fail(KT_ERR_UNSUPPORTED, "admit queue: the stored `used` of this engine is wider than int64 (kt_admit_sequential reads int64 tables)");
  if ((double)n * (double)e->thr_rows_hi > 2147483648.0)
    return e->fail(KT_ERR_OUT_OF_RANGE, "admit queue: n x throttle_rows = %lld x %d exceeds 2^31 matrix bytes", (long long)n, e->thr_rows_hi);
  // (a) who affects whom, for the whole queue in parallel (statuses against the current reserved amounts)
  int32_t rc = check_launch_locked(e, n, pod_rows, on_equal, KT_CHECK_STATUS_MATRIX, s, /*allow_small=*/false);
  if (rc != KT_OK || n == 0 || e->thr_rows_hi == 0) return rc;
  // (b) the queue in order, one wave, reserved amounts in LDS
  const bool commit = (flags & KT_ADMIT_COMMIT) != 0;
  KT_HIP(e, e->d_admit.reserve(kt::admit_state_bytes(e->thr_rows_hi, e->D) + 64));
  static const bool force_global = getenv("KT_ADMIT_FORCE_GLOBAL") != nullptr;  // test hook: HBM-resident state
  if (!kt::launch_admit(e->pods, n, pod_rows ? e->d_rows.p : nullptr, e->tt, e->thr_rows_hi, e->D, on_equal != 0, commit,
                        e->d_status.p, e->d_summary.p, e->d_admit.p, force_global, s))
    return e->fail(KT_ERR_UNSUPPORTED, "admit queue: %d throttle rows exceed the kernel's LDS list", e->thr_rows_hi);
  KT_HIP(e, hipGetLastError());
  if (commit) {
    e->reserved_dev_newer = true;
    std::lock_guard<std::mutex> g(e->recs_mu);
    e->recs_valid = false;
  }
  return KT_OK;
}

int32_t kt_fetch_reserved(kt_engine* e, int32_t n, const int32_t* rows, const kt_amounts* out) {
  if (!e || !out || n < 0 || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  for (int32_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= e->thr_rows_hi) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", rows[i]);
  int32_t rc = sync_status_to_host(e);
  if (rc != KT_OK) return rc;
  for (int32_t i = 0; i < n; ++i) amount_to_table(e->thr[(size_t)rows[i]].reserved, *out, (size_t)i, e->D);
  return KT_OK;
}

static int32_t check_fetch_locked(kt_engine* e, int64_t n, uint64_t* out_summary, uint8_t* out_status);

int32_t kt_check_fetch(kt_engine* e, int64_t n, uint64_t* out_summary, uint8_t* out_status) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  return check_fetch_locked(e, n, out_summary, out_status);
}

// ---- the few-pod path: what the scheduler's PreFilter actually calls (one pod per call, plugin.go:148-215)
static int32_t few_setup(kt_engine* e) {  // under the exclusive lock
  if (e->few_ready) return KT_OK;
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));  // CheckRecs written so far carry no event: let them land
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
  KT_HIP(e, hipStreamCreateWithPriority(&e->small_stream, hipStreamNonBlocking, hi));
  for (auto& ev : e->recs_ev) KT_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  KT_HIP(e, e->d_few_acc.reserve(8));
  KT_HIP(e, e->d_few_ticket.reserve(4));
  KT_HIP(e, hipMemsetAsync(e->d_few_acc.p, 0, 8 * 8, e->small_stream));
  KT_HIP(e, hipMemsetAsync(e->d_few_ticket.p, 0, 4 * 4, e->small_stream));
  KT_HIP(e, hipHostMalloc((void**)&e->h_few, 16 * 8, hipHostMallocMapped));
  memset(e->h_few, 0, 16 * 8);
  KT_HIP(e, hipStreamSynchronize(e->small_stream));
  e->few_ready = true;
  return KT_OK;
}

static inline bool few_shape_ok(const kt_engine* e, int64_t n, const int64_t* pod_rows, const uint64_t* out_summary, const uint8_t* out_status) {
  static const bool disabled = getenv("KT_NO_FEW") != nullptr;  // A/B runs: every kt_check through the staged small launch
  return !disabled && n >= 1 && n <= 8 && pod_rows && out_summary && !out_status && e->cfg.kernel_variant == 0;
}

// Under the SHARED lock (+ small_mu): nothing of the engine's host state is modified except the fields only this path
// touches.  Returns 1 when served, 0 when the caller has to take the exclusive path, < 0 on error.
static int32_t check_few_shared(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint64_t* out_summary) {
  if (!e->few_ready || e->program_dirty || e->status_host_dirty || e->hindex.has_slow || e->dindex.has_long || e->dindex.n_slow != 0 || e->n_overflow != 0 ||
      e->thr_rows_hi <= 0 || e->dindex.n_chunks == 0)
    return 0;
  for (int64_t i = 0; i < n; ++i)
    if (pod_rows[i] < 0 || pod_rows[i] >= e->cfg.pod_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "pod row %lld", (long long)pod_rows[i]);
  // which generation of CheckRecs: the current one once the kernel that writes it has completed, else the previous one
  int b;
  bool wait_cur = false;
  {
    std::lock_guard<std::mutex> g(e->recs_mu);
    if (!e->recs_valid || e->recs_eq != (on_equal != 0) || e->recs_DT != kt::dt_bucket_ix(e->D)) return 0;
    b = e->recs_cur;
    if (e->recs_ev_pending[b] && hipEventQuery(e->recs_ev[b]) == hipSuccess) e->recs_ev_pending[b] = false;
    if (e->recs_ev_pending[b]) {
      const int pb = 1 - b;
      if (e->recs_prev_valid && e->recs_ev_pending[pb] && hipEventQuery(e->recs_ev[pb]) == hipSuccess) e->recs_ev_pending[pb] = false;
      if (e->recs_prev_valid && !e->recs_ev_pending[pb]) b = pb;
      else wait_cur = true;  // two reconciles in flight: wait for the newer one
    }
  }
  if (wait_cur) KT_HIP(e, hipStreamWaitEvent(e->small_stream, e->recs_ev[b], 0));
  order_behind_ingest(e, e->small_stream);  // (a pod event right before this PreFilter: behind its kernel on the device)
  const uint64_t seq = ++e->few_seq;
  if (!kt::launch_check_few(e->pods, (int)n, pod_rows, e->sp, e->dindex, e->d_recs2[b].p, e->d_few_acc.p, e->d_few_ticket.p, e->h_few, e->h_few + 8,
                            seq, e->small_stream))
    return 0;
  KT_HIP(e, hipGetLastError());
  // the last workgroup writes the words and then the sequence number into pinned memory: spin on it
  volatile uint64_t* seqp = (volatile uint64_t*)(e->h_few + 8);
  bool done = false;
  for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
    if (*seqp == seq) {
      done = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!done) {  // far beyond any plausible latency: let the runtime report what happened
    KT_HIP(e, hipStreamSynchronize(e->small_stream));
    if (*seqp != seq) return e->fail(KT_ERR_DEVICE, "kt_check_few: no completion signal");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int64_t i = 0; i < n; ++i) out_summary[i] = e->h_few[i];
  e->few_served.fetch_add(1, std::memory_order_relaxed);
  return 1;
}

// launch + fetch as ONE critical section: what a caller needs when other threads use the engine at the same time
// (Unreserve from binding goroutines, reconcile workers) — a kt_check_launch / kt_check_fetch pair can be interleaved
int32_t kt_check(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint64_t* out_summary, uint8_t* out_status) {
  if (!e || n < 0) return KT_ERR_INVALID_ARGUMENT;
  const bool few = few_shape_ok(e, n, pod_rows, out_summary, out_status);
  if (few) {
    std::shared_lock<std::shared_mutex> rd(e->mu);
    std::lock_guard<std::mutex> sl(e->small_mu);
    KT_HIP(e, hipSetDevice(e->device));
    settle_ingest(e);  // a pod event fed just before: PreFilter sees it
    const int32_t rc = check_few_shared(e, n, pod_rows, on_equal, out_summary);
    if (rc != 0) return rc < 0 ? rc : KT_OK;
  }
  LaunchLock lk(e, few && !e->few_ready);  // the one-time set-up of the few-pod path changes what those checks read
  KT_HIP(e, hipSetDevice(e->device));
  if (few && !e->few_ready) {
    int32_t rc0 = few_setup(e);
    if (rc0 != KT_OK) return rc0;
  }
  int32_t rc = check_launch_locked(e, n, pod_rows, on_equal, out_status ? KT_CHECK_STATUS_MATRIX : 0u, e->own_stream);
  if (rc != KT_OK) return rc;
  return check_fetch_locked(e, n, out_summary, out_status);
}

static int32_t check_fetch_locked(kt_engine* e, int64_t n, uint64_t* out_summary, uint8_t* out_status) {
  if (!e->check_ready) return e->fail(KT_ERR_NOT_READY, "kt_check_fetch before kt_check_launch");
  if (n < 0 || n > e->check_n) return e->fail(KT_ERR_OUT_OF_RANGE, "n=%lld, last check had %lld pods", (long long)n, (long long)e->check_n);
  if (out_status && !e->check_has_status) return e->fail(KT_ERR_NOT_READY, "status matrix was not requested at launch");
  hipStream_t s = e->last_stream ? e->last_stream : e->own_stream;
  const bool from_pinned = e->check_in_h_small && e->h_small;  // the kernel already wrote the words to host memory
  if (n && out_summary && !from_pinned) KT_HIP(e, hipMemcpyAsync(out_summary, e->d_summary.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (n && out_status && e->check_T)  // the matrix was written with the row stride in effect at launch
    KT_HIP(e, hipMemcpyAsync(out_status, e->d_status.p, (size_t)n * (size_t)e->check_T, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));
  if (n && out_summary && from_pinned) memcpy(out_summary, e->h_small, (size_t)n * 8);
  return KT_OK;
}

// affectedPods restricted to the pods a caller names (throttle_controller.go:221-246 / clusterthrottle_controller.go:224-270):
// for each of the n pod rows and each of the m throttle rows — does the throttle's selector (namespace side included) match
// the pod as the engine holds it NOW.  What unreserveAffectedPods (throttle_controller.go:135-155) needs: a reservation is
// released behind a reconcile only for a pod that is IN the reconciled throttle's affected set — one whose labels moved on
// after Reserve is not.  One status-matrix check of those rows (a small launch), read by column.
int32_t kt_affected_pods(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t m, const int32_t* throttle_rows, uint8_t* out) {
  if (!e || n < 0 || m < 0 || (n > 0 && !pod_rows) || (m > 0 && !throttle_rows) || (n > 0 && m > 0 && !out)) return KT_ERR_INVALID_ARGUMENT;
  if (n == 0 || m == 0) return KT_OK;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  int32_t rc = ensure_ready(e, e->own_stream);
  if (rc != KT_OK) return rc;
  const int32_t T = e->thr_rows_hi;
  for (int32_t j = 0; j < m; ++j)
    if (throttle_rows[j] < 0 || throttle_rows[j] >= T) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", throttle_rows[j]);
  if ((rc = check_launch_locked(e, n, pod_rows, 0, KT_CHECK_STATUS_MATRIX, e->own_stream)) != KT_OK) return rc;
  std::vector<uint8_t> st((size_t)n * (size_t)T);
  if ((rc = check_fetch_locked(e, n, nullptr, st.data())) != KT_OK) return rc;
  // the call used the engine's ONE check slot: a kt_check_launch of the caller that was pending is gone — its kt_check_fetch must
  // answer KT_ERR_NOT_READY instead of handing out these pods' results
  e->check_ready = false;
  for (int64_t i = 0; i < n; ++i)
    for (int32_t j = 0; j < m; ++j) {
      const uint8_t v = st[(size_t)i * (size_t)T + (size_t)throttle_rows[j]];
      out[(size_t)i * (size_t)m + (size_t)j] = v == KT_STATUS_ERROR ? (uint8_t)KT_STATUS_ERROR : v != KT_STATUS_NOT_AFFECTED ? 1 : 0;
    }
  return KT_OK;
}


int32_t kt_paged_check(kt_engine* const* pages, int32_t n_pages, int64_t n, const int64_t* pod_rows, int32_t on_equal,
                       uint64_t* out_summary, uint8_t* out_status) {
  if (!pages || n_pages < 1 || n < 0) return KT_ERR_INVALID_ARGUMENT;
  for (int32_t k = 0; k < n_pages; ++k)
    if (!pages[k]) return KT_ERR_INVALID_ARGUMENT;
  int32_t T = 0;
  int32_t rc = kt_throttle_rows(pages[0], &T);
  if (rc != KT_OK) return rc;
  for (int32_t k = 1; k < n_pages; ++k) {
    int32_t Tk = 0;
    if ((rc = kt_throttle_rows(pages[k], &Tk)) != KT_OK) return rc;
    if (Tk != T) return pages[k]->fail(KT_ERR_INVALID_ARGUMENT, "page %d holds %d throttle rows, page 0 %d: every page holds every throttle", k, Tk, T);
  }
  if (n == 0) return KT_OK;
  // CheckThrottleStatus precedence (first hit wins, throttle_types.go:128-153): exceeds > active > insufficient > not throttled
  auto rank = [](uint8_t v) -> uint8_t {
    return v == KT_STATUS_ERROR ? 5 : v == KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD ? 4 : v == KT_STATUS_ACTIVE ? 3 : v == KT_STATUS_INSUFFICIENT ? 2 : v == KT_STATUS_NOT_THROTTLED ? 1 : 0;
  };
  static const uint8_t code[6] = {KT_STATUS_NOT_AFFECTED, KT_STATUS_NOT_THROTTLED, KT_STATUS_INSUFFICIENT, KT_STATUS_ACTIVE, KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD, KT_STATUS_ERROR};
  const size_t cells = (size_t)n * (size_t)(T > 0 ? T : 1);
  std::vector<uint8_t> acc(cells, 0), page(cells);
  for (int32_t k = 0; k < n_pages; ++k) {
    if ((rc = kt_check(pages[k], n, pod_rows, on_equal, nullptr, page.data())) != KT_OK) return rc;
    for (size_t i = 0; i < (size_t)n * (size_t)T; ++i) acc[i] = std::max(acc[i], rank(page[i]));
  }
  for (int64_t i = 0; i < n; ++i) {
    uint64_t n_exc = 0, n_act = 0, n_ins = 0;
    bool err = false;
    for (int32_t t = 0; t < T; ++t) {
      const uint8_t r = acc[(size_t)i * T + t];
      err |= r == 5, n_exc += r == 4, n_act += r == 3, n_ins += r == 2;
      if (out_status) out_status[(size_t)i * T + t] = code[r];
    }
    if (out_summary) out_summary[i] = err ? 2ull : ((n_exc | n_act | n_ins) ? 1ull : 0ull) | n_exc << 4 | n_act << 24 | n_ins << 44;
  }
  return KT_OK;
}

int32_t kt_paged_reconcile(kt_engine* const* pages, int32_t n_pages, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t n,
                           const kt_status* page_out, uint8_t* replaced_any, uint8_t* error_any) {
  if (!pages || n_pages < 1 || n < 0 || !page_out) return KT_ERR_INVALID_ARGUMENT;
  for (int32_t k = 0; k < n_pages; ++k)
    if (!pages[k]) return KT_ERR_INVALID_ARGUMENT;
  // With APPLY a page that has reconciled holds the NEW status; a failure on a later page must not leave the pages disagreeing
  // about calculatedThreshold / throttled without the caller knowing which did what.  So: every page is first
  // reconciled WITHOUT apply (a dry run that can fail — LDS budgets, sums out of range — on any page before anything is stored),
  // and only when all succeeded is the step run again with the caller's flags; a failure in that second pass (device errors only:
  // the same launches just succeeded) still drains every launched page's result and is reported after the last page.
  int32_t rc;
  if (flags & KT_RECONCILE_APPLY)
    for (int32_t k = 0; k < n_pages; ++k)
      if ((rc = kt_reconcile_launch(pages[k], now_s, now_ns, flags & ~KT_RECONCILE_APPLY, nullptr)) != KT_OK) return rc;
  int32_t launched = 0, first_err = KT_OK;
  for (; launched < n_pages; ++launched)  // (enqueued on every page's own stream: the pages run side by side on the device)
    if ((rc = kt_reconcile_launch(pages[launched], now_s, now_ns, flags, nullptr)) != KT_OK) { first_err = rc; break; }
  if (replaced_any) memset(replaced_any, 0, (size_t)n);
  if (error_any) memset(error_any, 0, (size_t)n);
  for (int32_t k = 0; k < launched; ++k) {
    if ((rc = kt_reconcile_fetch(pages[k], n, &page_out[k])) != KT_OK) {
      if (first_err == KT_OK) first_err = rc;
      continue;
    }
    for (int32_t i = 0; i < n; ++i) {
      if (replaced_any && page_out[k].calc_at_nonzero) replaced_any[i] |= page_out[k].calc_at_nonzero[i] != 0;
      if (error_any && page_out[k].error) error_any[i] |= page_out[k].error[i] != 0;
    }
  }
  return first_err;
}

