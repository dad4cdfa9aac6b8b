// kt_engine_reconcile.cpp — [Cluster]ThrottleController.reconcile, aggregation part (throttle_controller.go:103-133): the scan
// of the counted pods, the exchange between GPUs (kt_comm_*: RCCL), the finalize, and the fetches of their results.
#include "kt_engine_impl.h"

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.err = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) r.err = "librccl.so lacks the nccl* entry points";
  });
  return &r;
}
static constexpr int kNcclInt64 = 4, kNcclSum = 0;  // rccl.h: ncclInt64, ncclSum

// ---------------------------------------------------------------------------------------------------
// kt_comm_*: the reconcile's one exchange as a native RCCL all-reduce (no framework in the process).
// librccl.so is loaded on first use: an engine that never talks to another GPU does not depend on it.
// ---------------------------------------------------------------------------------------------------
int32_t kt_comm_unique_id(void* out_id128) {
  if (!out_id128) return KT_ERR_INVALID_ARGUMENT;
  Rccl* r = rccl();
  if (!r->err.empty()) {
    g_create_error = r->err;
    return KT_ERR_UNSUPPORTED;
  }
  const int rc = r->GetUniqueId(out_id128);
  if (rc != 0) {
    g_create_error = std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error");
    return KT_ERR_DEVICE;
  }
  return KT_OK;
}

int32_t kt_comm_init(kt_engine* e, int32_t rank, int32_t world, const void* id128) {
  if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  Rccl* r = rccl();
  if (!r->err.empty()) return e->fail(KT_ERR_UNSUPPORTED, "%s", r->err.c_str());
  if (e->comm) return e->fail(KT_ERR_INVALID_ARGUMENT, "kt_comm_init: the engine already has a communicator");
  Rccl::Id128 id;
  memcpy(id.b, id128, sizeof id.b);
  const int rc = r->CommInitRank(&e->comm, world, id, rank);
  if (rc != 0) {
    e->comm = nullptr;
    return e->fail(KT_ERR_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString ? r->GetErrorString(rc) : "error");
  }
  e->comm_rank = rank, e->comm_world = world;
  if (world > e->exchange_world) {
    e->exchange_world = world;
    if (world > 4) e->req_sums_valid = false;  // the per-rank bound shrinks: count again at the next reconcile
  }
  return KT_OK;
}

// the number of ranks whose partials the caller sums between kt_aggregate_launch and kt_finalize_launch with its OWN
// collective (kt_partial_used_buffer / kt_use_partial_buffer); kt_comm_init sets it by itself
int32_t kt_set_exchange_world(kt_engine* e, int32_t world) {
  if (!e || world < 1) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  if (world != e->exchange_world && (world > 4 || e->exchange_world > 4 || e->wide)) e->req_sums_valid = false;  // (a wide engine decides again: kt_set_wide_sums)
  e->exchange_world = world;
  return KT_OK;
}

int32_t kt_set_wide_sums(kt_engine* e, int32_t mode) {
  if (!e || (mode != 0 && mode != 1)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  if (mode == 1 && e->incremental) return e->fail(KT_ERR_UNSUPPORTED, "kt_set_wide_sums(1): an incremental engine keeps int64 partials");
  if (mode != e->wide_mode) e->req_sums_valid = false;  // the next aggregate decides again
  e->wide_mode = mode;
  return KT_OK;
}

int32_t kt_partial_words(kt_engine* e, int64_t* n_int64, int32_t* wide) {
  if (!e || !n_int64) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  if (!e->agg_pending) return e->fail(KT_ERR_NOT_READY, "kt_partial_words: no partials pending (kt_aggregate_launch first)");
  *n_int64 = (int64_t)e->agg_words;
  if (wide) *wide = e->agg_wide ? 1 : 0;
  return KT_OK;
}

int32_t kt_comm_allreduce_partial(kt_engine* e, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (!e->comm) return e->fail(KT_ERR_NOT_READY, "kt_comm_allreduce_partial before kt_comm_init");
  hipStream_t s = pick_stream(e, stream);
  if (!e->agg_pending) return e->fail(KT_ERR_NOT_READY, "kt_comm_allreduce_partial: no partials pending (kt_aggregate_launch first)");
  KT_CHECK_PARTIALS_CURRENT(e, "kt_comm_allreduce_partial");
  const size_t words = e->agg_words;  // what the scan filled, not what the throttle table holds now
  if (!words) return KT_OK;
  if (!e->partial()) return e->fail(KT_ERR_NOT_READY, "no partial buffer yet: kt_aggregate_launch first");
  if (e->ext_partial && (int64_t)words > e->ext_partial_words)
    return e->fail(KT_ERR_OUT_OF_RANGE, "caller partial buffer holds %lld words, %lld needed", (long long)e->ext_partial_words,
                   (long long)words);
  Rccl* r = rccl();
  const int rc = r->AllReduce(e->partial(), e->partial(), words, kNcclInt64, kNcclSum, e->comm, s);
  if (rc != 0) return e->fail(KT_ERR_DEVICE, "ncclAllReduce: %s", r->GetErrorString ? r->GetErrorString(rc) : "error");
  e->last_stream = s;
  return KT_OK;
}

int32_t kt_comm_destroy(kt_engine* e) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (!e->comm) return KT_OK;
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  (void)rccl()->CommDestroy(e->comm);
  e->comm = nullptr;
  e->comm_world = 1, e->comm_rank = 0;
  return KT_OK;
}


// ---------------------------------------------------------------------------------------------------
// reconcile
// ---------------------------------------------------------------------------------------------------

// every aggregate launch gets an epoch; a workgroup stamps the slabs it spills with it (kt_reduce_bitmap_slabs then
// leaves alone what a namespace-ordered scan did not write)
int32_t slab_tags(kt_engine* e, kt::AggScan& sc, hipStream_t s) {
  const size_t need = (size_t)e->dindex.n_chunks * kt::kSlabTagStride + 1;
  if (e->d_slab_tag.cap < need) {
    if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
    KT_HIP(e, e->d_slab_tag.reserve(need));
    KT_HIP(e, hipMemsetAsync(e->d_slab_tag.p, 0, e->d_slab_tag.cap * 4, s));
    e->slab_epoch = 0;
  }
  if (++e->slab_epoch == 0u) {  // wrapped: start over with clean tags
    KT_HIP(e, hipMemsetAsync(e->d_slab_tag.p, 0, e->d_slab_tag.cap * 4, s));
    e->slab_epoch = 1;
  }
  sc.slab_tag = e->d_slab_tag.p, sc.epoch = e->slab_epoch;
  return KT_OK;
}

// resource.Quantity never overflows (it promotes to big decimals); the engine's exact range is int64.  Every sum a
// scan, a delta scan or the exchange between GPUs forms is a sum over some of the CURRENT pods, so one exact total of
// |request| per dimension proves all of them in range — or names the dimension that needs a coarser scale.  The host
// keeps an upper bound of that total (req_sum_bound: it only grows with what is fed); the exact count on the device runs
// when the bound passes 2^60 and resets it.  2^60 per GPU leaves the headroom for up to 8 ranks' partials to meet in an
// all-reduce.
int32_t request_sums_in_range(kt_engine* e, hipStream_t s) {
  if (e->req_sums_valid && !(e->wide_mode == 1 && !e->wide)) return KT_OK;
  KT_HIP(e, e->d_req_sums.reserve(32));
  kt::launch_sum_abs_requests(e->pods, e->pod_rows_hi, e->d_req_sums.p, s);
  KT_HIP(e, hipGetLastError());
  unsigned long long h[32];
  KT_HIP(e, hipMemcpyAsync(h, e->d_req_sums.p, sizeof(h), hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));
  bool wide = false;
  for (int d = 0; d < e->D; ++d) {
    const unsigned __int128 total = (unsigned __int128)h[2 * d] + ((unsigned __int128)h[2 * d + 1] << 32);
    e->req_sum_bound[d] = total;
    if (total > rank_sum_bound(e->exchange_world)) {
      // where the reference would promote to big decimals (resourcelist.go:48-54): two limb sums per dimension, joined in
      // 128 bits by kt_finalize — for engines that rescan (the maintained partials of an incremental engine are int64)
      if (e->incremental)
        return e->fail(KT_ERR_OVERFLOW_RISK,
                       "dimension %d: the requests of the pods held here add up beyond 2^60 at this scale (the reference would "
                       "promote to big decimals); an incremental engine needs a coarser scale for it", d);
      if ((unsigned __int128)e->pod_rows_hi * (unsigned __int128)e->exchange_world > ((unsigned __int128)1 << 30))
        return e->fail(KT_ERR_OVERFLOW_RISK, "dimension %d: wide sums hold for up to 2^30 pods over all ranks", d);
      // Several ranks: the layout of the exchanged buffer (one block of int64 sums, or two blocks of limb sums) must be
      // the SAME on every rank, and this total is a local fact — another rank's shard may well stay below the bound.  So
      // a rank never goes wide by itself: the caller switches every rank with kt_set_wide_sums(e, 1).
      if (e->exchange_world > 1 && e->wide_mode != 1)
        return e->fail(KT_ERR_OVERFLOW_RISK,
                       "dimension %d: the requests of this rank's pods add up beyond the exact range of one int64 block; with %d ranks "
                       "the two-block form must be agreed: call kt_set_wide_sums(e, 1) on every rank", d, e->exchange_world);
      wide = true;
    }
  }
  if (e->wide_mode == 1 && !wide) {
    if (e->incremental) return e->fail(KT_ERR_UNSUPPORTED, "kt_set_wide_sums(1): an incremental engine keeps int64 partials");
    if ((unsigned __int128)e->pod_rows_hi * (unsigned __int128)e->exchange_world > ((unsigned __int128)1 << 30))
      return e->fail(KT_ERR_OVERFLOW_RISK, "wide sums hold for up to 2^30 pods over all ranks");
    wide = true;
  }
  if (wide != e->wide) e->countable_valid = false;  // packed request words only exist for sums inside int64
  e->wide = wide;
  e->req_sums_valid = true;
  return KT_OK;
}

static void upgrade_launch_lock(kt_engine* e) {
  if (e->cur_launch_lock) ((LaunchLock*)e->cur_launch_lock)->upgrade();
}

int32_t aggregate_locked(kt_engine* e, hipStream_t s, bool allow_fused) {
  e->fused_pending = false;
  int32_t rc = ensure_ready(e, s);
  if (rc != KT_OK) return rc;
  if ((rc = request_sums_in_range(e, s)) != KT_OK) return rc;
  const size_t block_words = (size_t)e->thr_rows_hi * kt::partial_stride(e->D);
  const size_t words = block_words * (e->wide ? 2 : 1);  // wide: the low-limb sums, then the high-part sums
  e->agg_wide = e->wide;
  if (e->ext_partial && (int64_t)words > e->ext_partial_words)
    return e->fail(KT_ERR_OUT_OF_RANGE, "caller partial buffer holds %lld words, %lld needed",
                   (long long)e->ext_partial_words, (long long)words);
  if (e->incremental && e->agg_valid) {
    // the partials were kept current by the pod event path: no scan
    if (words) KT_HIP(e, hipMemcpyAsync(e->partial(), e->d_agg.p, words * 8, hipMemcpyDeviceToDevice, s));
    e->last_kernel[KT_KERNEL_AGGREGATE] = "(incremental: no scan)";
    e->last_stream = s;
    e->agg_pending = true, e->agg_words = words, e->agg_gen = e->program_gen;
    return KT_OK;
  }
  // a multi-chunk index is scanned in namespace order (tiles share their word lists, workgroups skip foreign chunks)
  const bool by_ns = (e->dindex.n_chunks > 1 || e->sw[kSw_FORCE_NS_ORDER]) && !e->sw[kSw_NO_NS_ORDER];
  if ((rc = settle_view_patches(e, s)) != KT_OK) return rc;
  // pod events appended records behind the listed ones; a scan that will gather through the row list instead of streaming
  // the view cannot tell them from the list's zeroed padding: list again
  if (e->countable_valid && e->view_extra && !(!e->sw[kSw_NO_SCAN_VIEW] && (by_ns || e->dindex.n_chunks == 1))) e->countable_valid = false;
  if (e->cfg.kernel_variant != 1 && (!e->countable_valid || e->countable_by_ns != by_ns)) {  // pods changed since the last scan: which rows does a reconcile look at
    if (e->last_stream && e->last_stream != s) KT_HIP(e, hipStreamSynchronize(e->last_stream));
    KT_HIP(e, e->d_countable.reserve((size_t)e->cfg.pod_capacity + 1));
    KT_HIP(e, e->d_n_countable.reserve(1));
    if (by_ns) {
      KT_HIP(e, e->d_ns_cursor.reserve((size_t)e->sp.n_ns + 1));
      kt::launch_order_rows_by_ns(e->pods, e->pod_rows_hi, /*countable_only=*/true, (uint32_t)e->sp.n_ns,
                                  e->d_ns_cursor.p, e->d_countable.p, e->d_n_countable.p, s);
    } else {
      KT_HIP(e, hipMemsetAsync(e->d_n_countable.p, 0, 8, s));
      kt::launch_compact_countable(e->pods, e->pod_rows_hi, e->d_countable.p, e->d_n_countable.p, s);
    }
    e->countable_by_ns = by_ns;
    KT_HIP(e, hipGetLastError());
    KT_HIP(e, hipMemcpyAsync(&e->n_countable, e->d_n_countable.p, 8, hipMemcpyDeviceToHost, s));
    const bool plan_ranges = by_ns && !e->sw[kSw_NO_WG_RANGES];
    if (plan_ranges) {  // the ends of the namespaces' records travel with the row count: the ranges are planned on the host
      e->h_ns_end.resize((size_t)e->sp.n_ns + 1);
      KT_HIP(e, hipMemcpyAsync(e->h_ns_end.data(), e->d_ns_cursor.p, (size_t)e->sp.n_ns * 8, hipMemcpyDeviceToHost, s));
    }
    KT_HIP(e, hipStreamSynchronize(s));
    e->range_c_G = 0;
    if (plan_ranges && e->n_countable > 0) {
      e->range_c_G = kt::aggregate_blocks((int64_t)e->n_countable);
      KT_HIP(e, e->d_range_c.reserve((size_t)e->range_c_G + 2));
      e->h_range.resize((size_t)e->range_c_G + 2);
      kt::plan_wg_ranges(e->h_ns_end.data(), (uint32_t)e->sp.n_ns, (int64_t)e->n_countable, e->range_c_G, e->h_range.data());
      KT_HIP(e, hipMemcpyAsync(e->d_range_c.p, e->h_range.data(), e->h_range.size() * 4, hipMemcpyHostToDevice, s));
      KT_HIP(e, hipStreamSynchronize(s));  // (1 KB; h_range is reused)
    }
    e->pack = kt::PackPlan();
    // room for the pods that become countable before the next rebuild (kt_patch_scan_views appends them)
    const int64_t headroom = std::min<int64_t>(std::max<int64_t>(65536, (int64_t)e->n_countable / 16), e->cfg.pod_capacity - (int64_t)e->n_countable);
    // will the full scans stream the scan view?  (The packed fold exists in that form only: a scan that gathers through the row
    // list — KT_NO_SCAN_VIEW, KT_NO_NS_ORDER on a multi-chunk index — folds plain records.)
    const bool view_scan = !e->sw[kSw_NO_SCAN_VIEW] && (e->countable_by_ns || e->dindex.n_chunks == 1);
    // packed fold (PackPlan, kt_index.h) when every request of this engine is non-negative and the fields fit: sized
    // for the pods ONE workgroup scans with one workgroup per CU (two per CU scan fewer)
    if (view_scan && !e->incremental && !e->wide && !e->sw[kSw_NO_PACK] && !e->dindex.has_long) {
      const int64_t cap = (int64_t)e->n_countable + headroom;
      uint64_t slab_pods = kt::aggregate_slab_pods(cap, kt::aggregate_blocks(cap));
      // (planned ranges hold up to wg_range_cap records)
      if (e->range_c_G) slab_pods = std::max<uint64_t>(slab_pods, (uint64_t)kt::wg_range_cap((int64_t)e->n_countable, e->range_c_G) + 64u);
      e->pack = kt::make_pack_plan(e->D, e->max_abs, e->or_abs, e->neg_seen, slab_pods, /*pad_odd=*/true, kt::pack_max_words(e->D));
      if (e->pack.nw && e->pack.rec_bytes > e->dindex.cut_thr_bytes) e->pack = kt::PackPlan();  // the slab areas hold records of that size
    }
    if (!e->pack.nw && kt::agg_rec_bytes(e->D, e->incremental) > e->dindex.cut_thr_bytes) {
      // the plain fold is coming and the chunks were cut for the packed fold's records: cut again, for plain ones (once —
      // the engine then stays with plain-sized chunks), and start over on the new index
      upgrade_launch_lock(e);
      e->cut_plain = true, e->program_dirty = true;
      e->countable_valid = false;
      return aggregate_locked(e, s, allow_fused);
    }
    if (!e->sw[kSw_NO_SCAN_VIEW]) {
      // scan-ordered copies of the listed pods' records: the scan streams them instead of gathering through the list
      // (namespace order for a multi-chunk index, ascending rows otherwise)
      e->view_cap_c = (int64_t)e->n_countable + headroom;
      e->view_extra = 0;
      const size_t nc = (size_t)e->view_cap_c + 1;
      KT_HIP(e, e->d_vc_meta.reserve(nc));
      KT_HIP(e, e->d_vc_latom.reserve(nc * (size_t)e->pods.LA));
      if (e->pack.nw) KT_HIP(e, e->d_vc_pk.reserve(nc * (size_t)e->pack.stride));
      else KT_HIP(e, e->d_vc_req.reserve(nc * (size_t)e->pods.DS));
      KT_HIP(e, e->d_pos_c.reserve((size_t)e->cfg.pod_capacity + 1));
      KT_HIP(e, e->d_view_dirty.reserve(4));
      KT_HIP(e, hipMemsetAsync(e->d_pos_c.p, 0xFF, ((size_t)e->cfg.pod_capacity + 1) * 4, s));
      // the records past the listed ones are "no pod" until something is appended there
      KT_HIP(e, hipMemsetAsync(e->d_vc_meta.p + e->n_countable, 0, (size_t)(headroom + 1) * 8, s));
      KT_HIP(e, hipMemsetAsync(e->d_countable.p + e->n_countable, 0, (size_t)(headroom + 1) * 8, s));
      if (!e->view_check_dirty) KT_HIP(e, hipMemsetAsync(e->d_view_dirty.p, 0, 4, s));
      kt::launch_build_scan_view(e->pods, (int64_t)e->n_countable, e->d_countable.p, e->d_vc_meta.p, e->d_vc_latom.p,
                                 e->pack.nw ? nullptr : e->d_vc_req.p, s, e->pack.nw ? &e->pack : nullptr, e->d_vc_pk.p, e->d_pos_c.p);
      KT_HIP(e, hipGetLastError());
    }
    e->countable_valid = true;
  }
  if (words && e->clean_partial != (const void*)e->partial()) KT_HIP(e, hipMemsetAsync(e->partial(), 0, words * 8, s));
  e->clean_partial = nullptr;
  {
    TimedLaunch tl(e, KT_KERNEL_AGGREGATE, s);
    std::unique_ptr<TimedLaunch> tr;
    // reconcile in one call: the slab reduction of a packed scan is done by kt_reduce_finalize_packed
    // (single-chunk programs: with a chunked index most slabs are skipped and most throttles have several groups that meet
    // in the partial rows anyway — measured on the configs[4] shard: 111 us fused against 72 + 9 us)
    const bool defer = allow_fused && !e->incremental && !e->wide && e->dindex.n_chunks == 1 && !e->sw[kSw_NO_FUSED];
    auto after_scan = [&]() {  // the slab reduction is its own kernel: time it as its own family
      tl.stop_now();
      if (!(defer && e->pack.nw)) tr.reset(new TimedLaunch(e, KT_KERNEL_REDUCE, s));
    };
    // wide sums: two scans, the low 32-bit limb of every request into the first block, the rest into the second
    const int n_pass = e->wide ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
      const int limb = e->wide ? pass + 1 : 0;
      unsigned long long* target = e->partial() + (size_t)pass * block_words;
      if (e->cfg.kernel_variant == 1) {
        kt::launch_aggregate_dense(e->pods, e->pod_rows_hi, e->sp, e->uses_keys, target, s, limb);
        e->last_kernel[KT_KERNEL_AGGREGATE] = "kt_aggregate_dense";
      } else {
        kt::AggScan sc;
        sc.n = (int64_t)e->n_countable, sc.rows = e->d_countable.p, sc.counts = e->incremental, sc.nonneg = !e->neg_seen;
        sc.overflow_pods = e->n_overflow != 0;
        sc.small_window = e->sw[kSw_AGG_SMALL_WINDOW];
        sc.limb = limb;
        // contiguous tile ranges over the scan view; with a single chunk the order of the list does not matter
        sc.by_ns = !e->sw[kSw_NO_SCAN_VIEW] && (e->countable_by_ns || e->dindex.n_chunks == 1);
        // (records appended behind the listed ones by pod events exist in the VIEW only: a scan that gathers through the
        //  row list — KT_NO_NS_ORDER on a multi-chunk index — must not run over the list's zeroed padding = pod row 0)
        if (sc.by_ns) sc.n += e->view_extra;
        if (sc.by_ns) sc.v_meta = e->d_vc_meta.p, sc.v_latom = e->d_vc_latom.p, sc.v_req = e->pack.nw ? nullptr : e->d_vc_req.p;
        if (sc.by_ns && e->pack.nw) sc.pk = &e->pack, sc.v_pk = e->d_vc_pk.p;
        if (sc.by_ns && e->countable_by_ns && e->range_c_G) sc.wg_range = e->d_range_c.p, sc.wg_range_G = e->range_c_G;
        if ((rc = slab_tags(e, sc, s)) != KT_OK) return rc;
        sc.defer_reduce = defer && sc.pk != nullptr;
        const char* k = kt::launch_aggregate_indexed(e->pods, sc, e->sp, e->d_sp.p, e->dindex, target, e->d_slab.p, s,
                                                     pass == 0 ? std::function<void()>(after_scan) : std::function<void()>());
        if (!k) return e->fail(KT_ERR_UNSUPPORTED, "a chunk of the selector index exceeds the aggregate kernel's LDS budget (use kernel_variant 1)");
        e->last_kernel[KT_KERNEL_AGGREGATE] = k;
        e->ctr_packed_words.store(sc.launched_packed ? (int64_t)e->pack.nw : 0, std::memory_order_relaxed);
        if (sc.defer_reduce && sc.launched_packed) e->fused_pending = true, e->fused_nb = sc.launched_blocks, e->fused_epoch = sc.epoch, e->fused_pack = e->pack;
        e->last_kernel[KT_KERNEL_REDUCE] = e->fused_pending ? "(in kt_reduce_finalize_packed)" : sc.launched_packed ? "kt_reduce_packed_slabs" : "kt_reduce_bitmap_slabs";
      }
    }
  }
  KT_HIP(e, hipGetLastError());
  if (e->incremental) {  // baseline for the delta scans of the pod event path
    KT_HIP(e, e->d_agg.reserve(words + 1));
    if (words) KT_HIP(e, hipMemcpyAsync(e->d_agg.p, e->partial(), words * 8, hipMemcpyDeviceToDevice, s));
    e->agg_valid = true;
  }
  e->last_stream = s;
  e->agg_pending = true, e->agg_words = words, e->agg_gen = e->program_gen;
  return KT_OK;
}

// Pod event path of an incremental engine: the contribution of `n` pod rows (device list `rows_dev`,
// or the contiguous range row0 + [0, n)) is removed from (sign -1) or added to (+1) the maintained partials with one
// delta scan — the symmetric difference of throttle_controller.go:469-500 falls out of "remove the old pod, add the new".
int32_t delta_scan(kt_engine* e, int64_t n, const int64_t* rows_dev, int64_t row0, int sign, hipStream_t s) {
  if (!e->incremental || !e->agg_valid || n <= 0 || e->thr_rows_hi == 0) return KT_OK;
  kt::AggScan sc;
  sc.n = n, sc.rows = rows_dev, sc.row0 = row0, sc.counts = true, sc.sign = sign, sc.overflow_pods = e->n_overflow != 0;
  sc.small_window = e->sw[kSw_AGG_SMALL_WINDOW];
  int32_t rc = slab_tags(e, sc, s);
  if (rc != KT_OK) return rc;
  const char* k = kt::launch_aggregate_indexed(e->pods, sc, e->sp, e->d_sp.p, e->dindex, e->d_agg.p, e->d_slab.p, s, nullptr);
  if (!k) return e->fail(KT_ERR_UNSUPPORTED, "a chunk of the selector index exceeds the aggregate kernel's LDS budget");
  KT_HIP(e, hipGetLastError());
  return KT_OK;
}

// consume: kt_reconcile_launch — nobody reads the partials after this finalize, which leaves them zeroed for the next scan
int32_t finalize_locked(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, hipStream_t s, bool consume,
                               const uint8_t* row_mask) {
  KT_CHECK_PARTIALS_CURRENT(e, "kt_finalize_launch");
  int32_t rc = ensure_ready(e, s);
  if (rc != KT_OK) return rc;
  if (!e->agg_pending) e->agg_wide = e->wide;  // caller-provided partials, no aggregate of ours pending: the current mode's layout
  e->agg_pending = false;  // consumed (or caller-provided partials: nothing was pending)
  kt::ReconcileOut out{e->d_out_used.tab(), e->d_out_calc.tab(), e->d_out_used_hi.p, e->d_out_calc_updated.p, e->d_out_thrl_flag.p,
                       e->d_out_thrl_has.p, e->d_out_thrl_pod.p, e->d_out_error.p, e->d_out_next_s.p, e->d_out_next_ns.p};
  const bool apply = (flags & KT_RECONCILE_APPLY) != 0;
  // with APPLY the stored status changes: leave the CheckRecs of the new status behind (kt_prepare_check fused in),
  // built for the isThrottledOnEqual value the last check used (PreFilter: false)
  const int rec_DT = e->cfg.kernel_variant == 1 ? kt::dt_bucket(e->D) : kt::dt_bucket_ix(e->D);
  // the new generation of CheckRecs goes into the OTHER buffer when the current one is worth keeping for concurrent
  // single-pod checks (valid records of the same shape); otherwise it is rewritten in place, behind the checks in flight
  bool keep_prev;
  int wbuf;
  {
    std::lock_guard<std::mutex> g(e->recs_mu);
    keep_prev = apply && e->recs_valid && e->recs_DT == rec_DT && e->few_ready;
    wbuf = keep_prev ? 1 - e->recs_cur : e->recs_cur;
  }
  if (apply && !keep_prev) recs_invalidate_and_drain(e);
  if (apply && keep_prev) {
    // wbuf is the PREVIOUS generation's buffer — exactly what a concurrent few-pod check reads while the current buffer's
    // event is pending.  No new check may pick it (recs_prev_valid = false under recs_mu; such a check then waits on the
    // current buffer's event) and the one in flight has to finish before the finalize below rewrites it.
    {
      std::lock_guard<std::mutex> g(e->recs_mu);
      e->recs_prev_valid = false;
    }
    if (e->few_ready) std::lock_guard<std::mutex> drain(e->small_mu);
  }
  {
    TimedLaunch tl(e, KT_KERNEL_FINALIZE, s);
    if (e->fused_pending) {
      kt::launch_reduce_finalize_packed(e->tt, e->sp, e->D, e->dindex, e->fused_pack, e->d_slab.p, e->fused_nb, e->d_slab_tag.p, e->fused_epoch, e->partial(),
                                        consume, now_s, now_ns, apply, out, apply ? e->d_recs2[wbuf].p : nullptr, rec_DT, e->recs_eq, req_bound(e), s,
                                        row_mask, e->dindex.n_slow != 0 || e->n_overflow != 0);
      e->last_kernel[KT_KERNEL_FINALIZE] = "kt_reduce_finalize_packed";
    } else {
      kt::launch_finalize(e->tt, e->sp, e->D, e->partial(), consume, now_s, now_ns, apply, out, apply ? e->d_recs2[wbuf].p : nullptr, rec_DT,
                          e->recs_eq, req_bound(e), s, row_mask,
                          e->agg_wide ? e->partial() + (size_t)e->thr_rows_hi * kt::partial_stride(e->D) : nullptr);
      e->last_kernel[KT_KERNEL_FINALIZE] = "kt_finalize";
    }
    e->fused_pending = false;
  }
  if (apply) {
    if (e->few_ready) KT_HIP(e, hipEventRecord(e->recs_ev[wbuf], s));
    std::lock_guard<std::mutex> g(e->recs_mu);
    e->recs_ev_pending[wbuf] = e->few_ready;
    e->recs_prev_valid = keep_prev;
    e->recs_cur = wbuf;
    ++e->recs_seq[wbuf];
    e->recs_valid = true;  // e->recs_eq unchanged
    e->recs_DT = rec_DT;
  }
  e->clean_partial = consume ? (const void*)e->partial() : nullptr;
  KT_HIP(e, hipGetLastError());
  if (apply) e->status_dev_newer = true;
  e->reconcile_ready = true;
  e->reconcile_T = e->thr_rows_hi;
  e->last_stream = s;
  return KT_OK;
}

int32_t kt_aggregate_launch(kt_engine* e, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  return aggregate_locked(e, pick_stream(e, stream));
}

int32_t kt_partial_used_buffer(kt_engine* e, void** device_ptr, int64_t* n_int64) {
  if (!e || !device_ptr || !n_int64) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  int32_t rc = ensure_ready(e, e->own_stream);
  if (rc != KT_OK) return rc;
  *device_ptr = e->partial();
  // the words of the PENDING aggregate when there is one (a pod batch or kt_set_wide_sums may flip `wide` at the next
  // kt_aggregate_launch: re-query after each aggregate, or use kt_partial_words)
  *n_int64 = e->agg_pending ? (int64_t)e->agg_words : (int64_t)e->thr_rows_hi * kt::partial_stride(e->D) * ((e->wide || e->wide_mode == 1) ? 2 : 1);
  return KT_OK;
}

int32_t kt_use_partial_buffer(kt_engine* e, void* device_ptr, int64_t n_int64) {
  if (!e || (device_ptr && n_int64 <= 0)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  e->ext_partial = (unsigned long long*)device_ptr;
  e->ext_partial_words = device_ptr ? n_int64 : 0;
  e->clean_partial = nullptr;
  return KT_OK;
}

int32_t kt_finalize_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  e->fused_pending = false;  // a finalize of its own reads the partial buffer (kt_aggregate_launch reduced the slabs into it)
  return finalize_locked(e, now_s, now_ns, flags, pick_stream(e, stream));
}

int32_t kt_reconcile_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  hipStream_t s = pick_stream(e, stream);
  int32_t rc = aggregate_locked(e, s, /*allow_fused=*/true);
  if (rc != KT_OK) return rc;
  return finalize_locked(e, now_s, now_ns, flags, s, /*consume=*/!e->incremental);
}

int32_t kt_reconcile_rows_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t n,
                                 const int32_t* throttle_rows, void* stream) {
  if (!e || n < 0 || (n > 0 && !throttle_rows)) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  hipStream_t s = pick_stream(e, stream);
  for (int32_t i = 0; i < n; ++i)
    if (throttle_rows[i] < 0 || throttle_rows[i] >= e->cfg.throttle_capacity)
      return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", throttle_rows[i]);
  int32_t rc = ensure_ready(e, s);
  if (rc != KT_OK) return rc;
  // the keys of this reconcile as a byte per throttle row; the other rows keep (and report) their stored status
  std::vector<uint8_t> mask((size_t)e->thr_rows_hi + 1, 0);
  for (int32_t i = 0; i < n; ++i) {
    // a key beyond the rows in use was never upserted: silently "reconciling" it would report a stored status nobody wrote
    if (throttle_rows[i] >= e->thr_rows_hi)
      return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d was never upserted (rows in use: %d)", throttle_rows[i], e->thr_rows_hi);
    mask[(size_t)throttle_rows[i]] = 1;
  }
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  KT_HIP(e, e->d_row_mask.reserve(mask.size()));
  KT_HIP(e, hipMemcpyAsync(e->d_row_mask.p, mask.data(), mask.size(), hipMemcpyHostToDevice, s));
  KT_HIP(e, hipStreamSynchronize(s));  // `mask` goes out of scope
  rc = aggregate_locked(e, s, /*allow_fused=*/true);
  if (rc != KT_OK) return rc;
  return finalize_locked(e, now_s, now_ns, flags, s, /*consume=*/!e->incremental, e->d_row_mask.p);
}

int32_t kt_reconcile_fetch(kt_engine* e, int32_t n, const kt_status* out) {
  if (!e || !out) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (!e->reconcile_ready) return e->fail(KT_ERR_NOT_READY, "kt_reconcile_fetch before a reconcile launch");
  if (n < 0 || n > e->reconcile_T) return e->fail(KT_ERR_OUT_OF_RANGE, "n=%d, throttle rows of the last reconcile=%d", n, e->reconcile_T);
  hipStream_t s = e->last_stream ? e->last_stream : e->own_stream;
  const size_t N = (size_t)n;
  const int D = e->D;
  if (N) {
#define DL(dst, src, bytes) if (dst) KT_HIP(e, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, s))
    DL(out->used.v, e->d_out_used.v.p, N * D * 8);
    DL(out->used.present, e->d_out_used.present.p, N * 4);
    DL(out->used.count, e->d_out_used.count.p, N * 8);
    DL(out->used.has_count, e->d_out_used.has_count.p, N);
    DL(out->calc.v, e->d_out_calc.v.p, N * D * 8);
    DL(out->calc.present, e->d_out_calc.present.p, N * 4);
    DL(out->calc.count, e->d_out_calc.count.p, N * 8);
    DL(out->calc.has_count, e->d_out_calc.has_count.p, N);
    DL(out->calc_at_nonzero, e->d_out_calc_updated.p, N);
    DL(out->thrl_flag, e->d_out_thrl_flag.p, N * 4);
    DL(out->thrl_has, e->d_out_thrl_has.p, N * 4);
    DL(out->thrl_pod, e->d_out_thrl_pod.p, N);
    DL(out->error, e->d_out_error.p, N);
#undef DL
  }
  KT_HIP(e, hipStreamSynchronize(s));
  return KT_OK;
}

// High 64 bits of the last reconcile's `used` values (rows [0, n) x n_dims): all of them the sign extension of
// kt_reconcile_fetch's used.v unless the requests of the pods held add up beyond int64 (resource.Quantity never overflows,
// resourcelist.go:48-54: the engine then sums 32-bit limbs and joins them in 128 bits) — out_any_wide says whether any differs
int32_t kt_reconcile_fetch_used_hi(kt_engine* e, int32_t n, int64_t* out_hi, int32_t* out_any_wide) {
  if (!e || !out_hi) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (!e->reconcile_ready) return e->fail(KT_ERR_NOT_READY, "kt_reconcile_fetch_used_hi before a reconcile launch");
  if (n < 0 || n > e->reconcile_T) return e->fail(KT_ERR_OUT_OF_RANGE, "n=%d, throttle rows of the last reconcile=%d", n, e->reconcile_T);
  hipStream_t s = e->last_stream ? e->last_stream : e->own_stream;
  const size_t N = (size_t)n * (size_t)e->D;
  std::vector<int64_t> lo(N + 1);
  if (N) {
    KT_HIP(e, hipMemcpyAsync(out_hi, e->d_out_used_hi.p, N * 8, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(lo.data(), e->d_out_used.v.p, N * 8, hipMemcpyDeviceToHost, s));
  }
  KT_HIP(e, hipStreamSynchronize(s));
  int32_t any = 0;
  for (size_t i = 0; i < N; ++i) any |= out_hi[i] != (lo[i] < 0 ? -1 : 0);
  if (out_any_wide) *out_any_wide = any;
  return KT_OK;
}

// NextOverrideHappensIn of the last reconcile, as instants (has = 0: nothing ahead / row not reconciled)
int32_t kt_reconcile_fetch_next_override(kt_engine* e, int32_t n, int64_t* next_s, int32_t* next_ns, uint8_t* has) {
  if (!e || !next_s || !next_ns || !has) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (!e->reconcile_ready) return e->fail(KT_ERR_NOT_READY, "kt_reconcile_fetch_next_override before a reconcile launch");
  if (n < 0 || n > e->reconcile_T) return e->fail(KT_ERR_OUT_OF_RANGE, "n=%d, throttle rows of the last reconcile=%d", n, e->reconcile_T);
  hipStream_t s = e->last_stream ? e->last_stream : e->own_stream;
  if (n) {
    KT_HIP(e, hipMemcpyAsync(next_s, e->d_out_next_s.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(next_ns, e->d_out_next_ns.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  }
  KT_HIP(e, hipStreamSynchronize(s));
  for (int32_t i = 0; i < n; ++i) {
    has[i] = next_s[i] != INT64_MAX;
    if (!has[i]) next_s[i] = 0, next_ns[i] = 0;
  }
  return KT_OK;
}

