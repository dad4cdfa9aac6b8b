// kt_engine_feed.cpp — the state feed of the C-ABI: what the informer event handlers push (namespaces, pods, throttles, stored
// status, reserved amounts, whole snapshots).  throttle_controller.go:400-536, clusterthrottle_controller.go:428-570.
#include "kt_engine_impl.h"

int32_t kt_upsert_namespaces(kt_engine* e, const kt_snapshot* b, const int32_t* rows) {
  if (!e || !b) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  for (int32_t i = 0; i < b->n_ns; ++i) {
    const int32_t row = rows ? rows[i] : i;
    if (row < 0 || row >= e->cfg.namespace_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "namespace row %d", row);
  }
  for (int32_t i = 0; i < b->n_ns; ++i) {
    HostNamespace& n = e->ns[(size_t)(rows ? rows[i] : i)];
    n.valid = b->ns_valid ? b->ns_valid[i] != 0 : true;
    n.labels.clear();
    for (uint32_t k = b->ns_label_off[i]; k < b->ns_label_off[i + 1]; ++k)
      n.labels.emplace_back(b->ns_label_key[k], b->ns_label_pair[k]);
    e->ns_rows_hi = std::max(e->ns_rows_hi, (rows ? rows[i] : i) + 1);
  }
  if (b->n_ns > 0) e->program_dirty = true, ++e->ns_gen;
  return KT_OK;
}

int32_t kt_delete_namespaces(kt_engine* e, int32_t n, const int32_t* rows) {
  if (!e || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  for (int32_t i = 0; i < n; ++i) {
    if (rows[i] < 0 || rows[i] >= e->cfg.namespace_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "namespace row %d", rows[i]);
    e->ns[(size_t)rows[i]] = HostNamespace();
  }
  if (n > 0) e->program_dirty = true, ++e->ns_gen;
  return KT_OK;
}


// ---- pod events applied to the scan views in place
constexpr int64_t kPatchBatchMax = 65536;
// can a batch of n pod rows (largest |request| per dimension batch_max, OR of the values batch_or, a negative value seen)
// be applied to the current views?  The packed request words only hold what their plan was proved for.
static bool views_patchable(const kt_engine* e, int64_t n, const unsigned __int128* batch_max, const uint64_t* batch_or, bool batch_neg) {
  if (e->incremental || e->cfg.kernel_variant != 0 || e->program_dirty || n > kPatchBatchMax) return false;
  if (!e->countable_valid && !e->order_all_valid) return false;  // nothing to patch: the next scan builds anyway
  if (e->sw[kSw_NO_VIEW_PATCH]) return false;
  if (e->countable_valid) {
    if (e->d_vc_meta.p == nullptr || e->d_pos_c.p == nullptr) return false;
    if (e->view_extra + n > e->view_cap_c - (int64_t)e->n_countable) return false;
    if (e->pack.nw) {
      if (batch_neg) return false;
      for (int d = 0; d < e->D; ++d) {
        if (batch_max[d] > e->max_abs[d]) return false;  // a field may be too narrow
        if (e->pack.shift[d] && (batch_or[d] & ((1ull << e->pack.shift[d]) - 1ull))) return false;  // fewer common trailing zeros
      }
    } else if (!e->neg_seen && !batch_neg) {
      // unpacked view of an engine that could pack: a rebuild decides again
    }
  }
  return true;
}
// the views a pod event batch of n rows has to be applied to (host bookkeeping included: call once per batch)
static kt::ViewPatch view_patch_of(kt_engine* e, int64_t n) {
  kt::ViewPatch v{};
  if (e->countable_valid) {
    v.vc_meta = e->d_vc_meta.p, v.vc_latom = e->d_vc_latom.p, v.vc_req = e->pack.nw ? nullptr : e->d_vc_req.p, v.vc_pk = e->pack.nw ? e->d_vc_pk.p : nullptr;
    v.vc_rows = e->d_countable.p, v.pos_c = e->d_pos_c.p, v.n_c = e->d_n_countable.p, v.cap_c = e->view_cap_c;
    v.by_ns = e->countable_by_ns ? 1u : 0u;
    v.pk = e->pack;
    if (!e->countable_by_ns) e->view_extra += n;  // at most n appended
  }
  if (e->order_all_valid) {
    v.va_meta = e->d_va_meta.p, v.va_latom = e->d_va_latom.p, v.pos_a = e->d_pos_a.p, v.rows_a = e->view_rows_a;
  }
  v.dirty = e->d_view_dirty.p;
  if ((e->countable_valid && e->countable_by_ns) || e->order_all_valid) e->view_check_dirty = true;
  return v;
}
static int32_t patch_views(kt_engine* e, int64_t n, const int64_t* rows_dev, int64_t row0, hipStream_t s) {
  const kt::ViewPatch v = view_patch_of(e, n);
  kt::launch_patch_scan_views(e->pods, n, rows_dev, row0, v, s);
  KT_HIP(e, hipGetLastError());
  return KT_OK;
}
// before a scan uses a namespace-ordered view that was patched: did an entry have to move?
int32_t settle_view_patches(kt_engine* e, hipStream_t s) {
  if (!e->view_check_dirty) return KT_OK;
  uint32_t dirty = 0;
  KT_HIP(e, hipMemcpyAsync(&dirty, e->d_view_dirty.p, 4, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));
  if (dirty) {
    e->countable_valid = false, e->order_all_valid = false;
    KT_HIP(e, hipMemsetAsync(e->d_view_dirty.p, 0, 4, s));
  }
  e->view_check_dirty = false;
  return KT_OK;
}

static int32_t upsert_pods_locked(kt_engine* e, const kt_snapshot* b, const int64_t* rows) {
  const int D = e->D;
  if (b->D != D) return e->fail(KT_ERR_INVALID_ARGUMENT, "batch D=%d, engine D=%d", b->D, D);
  const int64_t n = b->n_pods;
  if (n <= 0) return KT_OK;
  // ---- validation + overflow bound (host pass over the batch; the data is copied once, below)
  int64_t hi = e->pod_rows_hi, ns_hi = e->pod_ns_hi;
  unsigned __int128 batch_max[KT_MAX_DIMS] = {0}, batch_total[KT_MAX_DIMS] = {0};
  uint64_t batch_or[KT_MAX_DIMS] = {0};
  const bool neg_before = e->neg_seen;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t row = rows ? rows[i] : i;
    if (row < 0 || row >= e->cfg.pod_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "pod row %lld", (long long)row);
    if (b->pod_ns[i] >= (uint32_t)e->cfg.namespace_capacity)
      return e->fail(KT_ERR_OUT_OF_RANGE, "pod %lld: namespace id %u", (long long)i, b->pod_ns[i]);
    ns_hi = std::max(ns_hi, (int64_t)b->pod_ns[i] + 1);
    if (b->pod_label_off[i + 1] - b->pod_label_off[i] > (uint32_t)e->L)
      return e->fail(KT_ERR_OUT_OF_RANGE, "pod %lld has %u labels, engine keeps %d", (long long)i,
                     b->pod_label_off[i + 1] - b->pod_label_off[i], e->L);
    hi = std::max(hi, row + 1);
    unsigned __int128 sum[KT_MAX_DIMS] = {0};
    for (uint32_t k = b->pod_ctr_off[i]; k < b->pod_ctr_off[i + 1]; ++k)
      for (int d = 0; d < D; ++d)
        if ((b->ctr_present[k] >> d) & 1u) {
          sum[d] += uabs(b->ctr_req[(size_t)k * D + d]);
          batch_or[d] |= (uint64_t)uabs(b->ctr_req[(size_t)k * D + d]);
          if (b->ctr_req[(size_t)k * D + d] < 0) e->neg_seen = true;
        }
    if (b->pod_ovh_present[i] >> 31)
      for (int d = 0; d < D; ++d)
        if ((b->pod_ovh_present[i] >> d) & 1u) {
          sum[d] += uabs(b->pod_ovh[(size_t)i * D + d]);
          batch_or[d] |= (uint64_t)uabs(b->pod_ovh[(size_t)i * D + d]);
          if (b->pod_ovh[(size_t)i * D + d] < 0) e->neg_seen = true;
        }
    for (int d = 0; d < D; ++d) batch_max[d] = std::max(batch_max[d], sum[d]), batch_total[d] += sum[d];
  }
  // a single request beyond 2^60 is refused here; whether the requests of all pods still ADD UP inside the exact range
  // is checked against their actual sum when a reconcile scans them (request_sums_in_range)
  for (int d = 0; d < D; ++d)
    if (batch_max[d] > kSumBound)
      return e->fail(KT_ERR_OVERFLOW_RISK, "dimension %d: a pod's request exceeds 2^60 at this scale; use a coarser scale for it", d);
  // the scan lists / views: patched in place when the batch fits what they were built for, else rebuilt by the next scan
  const bool patch = views_patchable(e, n, batch_max, batch_or, e->neg_seen && !neg_before);
  for (int d = 0; d < D; ++d) {
    if (batch_max[d] > e->max_abs[d]) e->recs_valid = false;  // kRecTight was judged against the old bound
    e->max_abs[d] = std::max(batch_max[d], e->max_abs[d]);
    e->or_abs[d] |= batch_or[d];
  }
  if (!patch) {
    e->countable_valid = false;
    e->order_all_valid = false;
  }
  // the overflow guard's bound grows by what this batch brings; only when it passes 2^60 does the next reconcile count
  // exactly on the device (request_sums_in_range), which also forgets the overwritten and deleted pods again
  for (int d = 0; d < D; ++d) {
    e->req_sum_bound[d] += batch_total[d];
    if (e->req_sum_bound[d] > rank_sum_bound(e->exchange_world)) e->req_sums_valid = false;
  }
  e->pod_ns_hi = ns_hi;
  if ((size_t)ns_hi > e->ns_compiled) e->program_dirty = true;  // a namespace row the compiled program does not cover yet
  // ---- stage + ingest in chunks
  hipStream_t s = e->own_stream;
  const int64_t chunk = 1 << 20;
  // A batch may name a pod row more than once (coalesced informer events: Add, then Update of the same pod) and the LAST entry
  // must win, as if the events had arrived one by one.  The kernels below run one thread / wave per entry, so two entries of one
  // row inside one launch would race for the row (the unfused path wrote a torn mix of both).  A chunk therefore ends
  // where a row would repeat: the chunks are launched in stream order, every launch sees unique rows, the later entry
  // overwrites the earlier one — and an incremental engine's delta scans take the first entry out again before the second goes in.
  std::unordered_set<int64_t> seen_rows;
  auto unique_prefix = [&](int64_t c0, int64_t max_n) -> int64_t {
    if (!rows || max_n <= 1) return max_n;
    bool ascending = true;
    for (int64_t i = 1; i < max_n && ascending; ++i) ascending = rows[c0 + i] > rows[c0 + i - 1];
    if (ascending) return max_n;  // (the usual case: no table needed)
    seen_rows.clear();
    seen_rows.reserve((size_t)std::min<int64_t>(max_n, 1 << 16));
    for (int64_t i = 0; i < max_n; ++i)
      if (!seen_rows.insert(rows[c0 + i]).second) return i;
    return max_n;
  };
  for (int64_t c0 = 0, cn = 0; c0 < n; c0 += cn) {
    cn = unique_prefix(c0, std::min(chunk, n - c0));
    const uint32_t lb = b->pod_label_off[c0], le = b->pod_label_off[c0 + cn];
    const uint32_t kb = b->pod_ctr_off[c0], ke = b->pod_ctr_off[c0 + cn];
    // layout of the staging buffer (8-byte aligned sections)
    size_t off = 0;
    auto sect = [&](size_t bytes) { size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    const size_t o_rows = sect(rows ? cn * 8 : 0), o_ns = sect(cn * 4), o_fl = sect(cn * 4), o_lo = sect((cn + 1) * 4),
                 o_lk = sect((size_t)(le - lb) * 4), o_lp = sect((size_t)(le - lb) * 4), o_co = sect((cn + 1) * 4),
                 o_ci = sect(ke - kb), o_cp = sect((size_t)(ke - kb) * 4), o_cr = sect((size_t)(ke - kb) * 8 * D),
                 o_op = sect(cn * 4), o_ov = sect((size_t)cn * 8 * D);
    // an informer event or a coalesced handful of them (the whole batch fits one pinned slot): no device staging copy —
    // the kernels read the slot where it lies — and no stream synchronisation: an event behind the kernels, and
    // settle_ingest() in every entry point that is not a pod feed call
    const bool slot_path = n <= chunk && off + 16 <= kt_engine::kEvSlotBytes && !e->incremental && !e->sw[kSw_SYNC_INGEST];
    kt_engine::EvSlot* slot = nullptr;
    if (slot_path) {
      slot = &e->ev_slots[e->ev_next];
      e->ev_next = (e->ev_next + 1) % kt_engine::kEvSlots;
      if (!slot->h) {
        KT_HIP(e, hipHostMalloc((void**)&slot->h, kt_engine::kEvSlotBytes, hipHostMallocDefault));
        KT_HIP(e, hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
      }
      if (!e->h_overflow) {
      KT_HIP(e, hipHostMalloc((void**)&e->h_overflow, 64, hipHostMallocDefault));
      memset(e->h_overflow, 0, 64);  // (word 1 is the sequence number settle_ingest compares with)
    }
      if (slot->used) KT_HIP(e, hipEventSynchronize(slot->ev));  // (eight feed calls ago: long done)
    } else {
      settle_ingest(e);  // the staged path below synchronises anyway
      KT_HIP(e, e->d_stage.reserve(off + 16));
    }
    // ONE launch (kt_feed_small) for an event batch: the workgroup first pulls the whole slot over the link with all its
    // threads, the batch pointers name that device copy (KT_FEED_NO_STAGE=1: the kernel walks the slot over the link)
    const bool fused = slot_path && cn <= kt::kFeedSmallMax && !e->sw[kSw_NO_FEED_FUSION];
    // an informer event proper — a pod or a few: one wave per pod, the slot pulled into LDS (kt_feed_few); the batch
    // pointers are then byte offsets into the slot (KT_NO_FEED_FEW=1: kt_feed_small's thread per pod, A/B)
    const bool few = fused && cn <= kt::kFeedFewMax && off <= kt::kFeedFewSlotMax && !e->sw[kSw_NO_FEED_FEW];
    const bool dev_copy = fused && !few && !e->sw[kSw_FEED_NO_STAGE];
    if (dev_copy) KT_HIP(e, e->d_ev_stage.reserve(kt_engine::kEvSlotBytes));
    uint8_t* st = few ? (uint8_t*)nullptr : dev_copy ? e->d_ev_stage.p : slot_path ? slot->h : e->d_stage.p;
    // a small batch is packed in pinned host memory and crosses in ONE copy; a bulk load copies its sections straight
    // from the caller's arrays
    const bool packed = !slot_path && off <= kPinnedStageBytes;
    if (packed && !e->h_stage) KT_HIP(e, hipHostMalloc((void**)&e->h_stage, kPinnedStageBytes, hipHostMallocDefault));
#define CP(o, src, bytes)                                                                               \
  if ((bytes) > 0) {                                                                                    \
    if (slot_path) memcpy(slot->h + (o), (src), (bytes));                                               \
    else if (packed) memcpy(e->h_stage + (o), (src), (bytes));                                          \
    else KT_HIP(e, hipMemcpyAsync(st + (o), (src), (bytes), hipMemcpyHostToDevice, s));                 \
  }
    if (rows) CP(o_rows, rows + c0, (size_t)cn * 8);
    CP(o_ns, b->pod_ns + c0, (size_t)cn * 4);
    CP(o_fl, b->pod_flags + c0, (size_t)cn * 4);
    CP(o_lo, b->pod_label_off + c0, (size_t)(cn + 1) * 4);
    CP(o_lk, b->pod_label_key + lb, (size_t)(le - lb) * 4);
    CP(o_lp, b->pod_label_pair + lb, (size_t)(le - lb) * 4);
    CP(o_co, b->pod_ctr_off + c0, (size_t)(cn + 1) * 4);
    CP(o_ci, b->ctr_init + kb, (size_t)(ke - kb));
    CP(o_cp, b->ctr_present + kb, (size_t)(ke - kb) * 4);
    CP(o_cr, b->ctr_req + (size_t)kb * D, (size_t)(ke - kb) * 8 * D);
    CP(o_op, b->pod_ovh_present + c0, (size_t)cn * 4);
    CP(o_ov, b->pod_ovh + (size_t)c0 * D, (size_t)cn * 8 * D);
#undef CP
    if (packed) KT_HIP(e, hipMemcpyAsync(st, e->h_stage, off, hipMemcpyHostToDevice, s));
    kt::PodBatchDev pb{};
    pb.n = cn;
    pb.rows = rows ? (const int64_t*)(st + o_rows) : nullptr;
    pb.row0 = c0;
    pb.ns = (const uint32_t*)(st + o_ns);
    pb.flags = (const uint32_t*)(st + o_fl);
    pb.label_off = (const uint32_t*)(st + o_lo);
    pb.label_key = (const uint32_t*)(st + o_lk);
    pb.label_pair = (const uint32_t*)(st + o_lp);
    pb.label_base = lb;
    pb.ctr_off = (const uint32_t*)(st + o_co);
    pb.ctr_init = (const uint8_t*)(st + o_ci);
    pb.ctr_present = (const uint32_t*)(st + o_cp);
    pb.ctr_req = (const int64_t*)(st + o_cr);
    pb.ctr_base = kb;
    pb.ovh_present = (const uint32_t*)(st + o_op);
    pb.ovh = (const int64_t*)(st + o_ov);
    // incremental engines: out with the old content of these rows, in with the new (a row that is not valid yet /
    // any more contributes nothing either way)
    if (e->incremental && e->program_dirty) e->agg_valid = false;  // selectors changed: the next reconcile rescans
    if (fused) {
      // ONE launch: ingest + translate + view patch, the overflow counter straight into the pinned word
      const bool tr = !e->program_dirty && e->pods.latom;
      kt::ViewPatch v{};
      if (patch) v = view_patch_of(e, cn);
      const bool spin = !e->sw[kSw_INGEST_EVENT_WAIT];  // (A/B: wait on the event as the first form of this path did)
      const unsigned long long seq = ++e->ingest_seq;
      if (few)
        kt::launch_feed_few(e->pods, pb, rows != nullptr, e->dindex, e->d_overflow.p, tr, patch ? &v : nullptr, e->h_overflow, slot->h, (uint32_t)off,
                            spin ? e->h_overflow + 1 : nullptr, seq, s);
      else
        kt::launch_feed_small(e->pods, pb, e->dindex, e->d_overflow.p, tr, patch ? &v : nullptr, e->h_overflow, dev_copy ? slot->h : nullptr,
                              dev_copy ? e->d_ev_stage.p : nullptr, dev_copy ? (uint32_t)off : 0u, spin ? e->h_overflow + 1 : nullptr, seq, s);
      KT_HIP(e, hipGetLastError());
      if (tr) e->overflow_in_flight = true;
      KT_HIP(e, hipEventRecord(slot->ev, s));
      slot->used = true;
      std::lock_guard<std::mutex> g(e->ingest_mu);
      e->ingest_ev = slot->ev;
      e->ingest_stream = s;
      e->ingest_spin_seq = spin ? seq : 0ull;
      e->ingest_pending.store(true, std::memory_order_release);
      continue;
    }
    int32_t drc = delta_scan(e, cn, pb.rows, pb.row0, -1, s);
    if (drc != KT_OK) return drc;
    kt::launch_ingest_pods(e->pods, pb, s);
    KT_HIP(e, hipGetLastError());
    if (!e->program_dirty && e->pods.latom) {  // atom rows of the new content (a dirty program translates every row when compiled)
      kt::launch_translate_pods(e->pods, cn, pb.rows, pb.row0, e->dindex, e->d_overflow.p, s);
      KT_HIP(e, hipGetLastError());
      if (slot_path) {
        KT_HIP(e, hipMemcpyAsync(e->h_overflow, e->d_overflow.p, 8, hipMemcpyDeviceToHost, s));
        e->overflow_in_flight = true;
      } else {
        KT_HIP(e, hipMemcpyAsync(&e->n_overflow, e->d_overflow.p, 8, hipMemcpyDeviceToHost, s));
      }
    }
    if ((drc = delta_scan(e, cn, pb.rows, pb.row0, +1, s)) != KT_OK) return drc;
    if (patch && (drc = patch_views(e, cn, pb.rows, pb.row0, s)) != KT_OK) return drc;
    if (slot_path) {
      KT_HIP(e, hipEventRecord(slot->ev, s));
      slot->used = true;
      std::lock_guard<std::mutex> g(e->ingest_mu);
      e->ingest_ev = slot->ev;
      e->ingest_stream = s;
      e->ingest_spin_seq = 0ull;  // several kernels: the event says when the last one is done
      e->ingest_pending.store(true, std::memory_order_release);
    } else {
      KT_HIP(e, hipStreamSynchronize(s));  // staging buffer is reused by the next chunk
    }
  }
  e->pod_rows_hi = hi;
  e->last_stream = s;
  return KT_OK;
}

int32_t kt_upsert_pods(kt_engine* e, const kt_snapshot* b, const int64_t* rows) {
  if (!e || !b) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e, /*settle=*/false);  // pod feed calls pipeline on the engine's stream
  KT_HIP(e, hipSetDevice(e->device));
  // kernels of another stream may still read the pod tables; what is in flight on the engine's own stream is ordered
  // before this call's kernels by the stream itself
  if (e->last_stream && e->last_stream != e->own_stream) {
    settle_ingest(e);
    KT_HIP(e, hipStreamSynchronize(e->last_stream));
  }
  return upsert_pods_locked(e, b, rows);
}

int32_t kt_delete_pods(kt_engine* e, int64_t n, const int64_t* rows) {
  if (!e || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e, /*settle=*/false);  // pipelines with the other pod feed calls (see kt_upsert_pods)
  KT_HIP(e, hipSetDevice(e->device));
  for (int64_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= e->cfg.pod_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "pod row %lld", (long long)rows[i]);
  if (n <= 0) return KT_OK;
  const bool slot_path = (size_t)n * 8 <= kt_engine::kEvSlotBytes && !e->incremental && !e->sw[kSw_SYNC_INGEST];
  if (!slot_path || (e->last_stream && e->last_stream != e->own_stream)) {
    settle_ingest(e);
    if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  }
  kt_engine::EvSlot* slot = nullptr;
  const int64_t* rows_dev;
  if (slot_path) {
    slot = &e->ev_slots[e->ev_next];
    e->ev_next = (e->ev_next + 1) % kt_engine::kEvSlots;
    if (!slot->h) {
      KT_HIP(e, hipHostMalloc((void**)&slot->h, kt_engine::kEvSlotBytes, hipHostMallocDefault));
      KT_HIP(e, hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
    }
    if (slot->used) KT_HIP(e, hipEventSynchronize(slot->ev));
    memcpy(slot->h, rows, (size_t)n * 8);
    rows_dev = (const int64_t*)slot->h;
  } else {
    KT_HIP(e, e->d_rows.reserve((size_t)n));
    KT_HIP(e, hipMemcpyAsync(e->d_rows.p, rows, (size_t)n * 8, hipMemcpyHostToDevice, e->own_stream));
    rows_dev = e->d_rows.p;
  }
  const unsigned __int128 no_max[KT_MAX_DIMS] = {0};
  const uint64_t no_or[KT_MAX_DIMS] = {0};
  const bool patch = views_patchable(e, n, no_max, no_or, false);
  if (!patch) {
    e->countable_valid = false;
    e->order_all_valid = false;
  }
  if (e->incremental && e->program_dirty) e->agg_valid = false;
  unsigned long long spin_seq = 0ull;
  if (slot_path && n <= kt::kFeedSmallMax && !e->sw[kSw_NO_FEED_FUSION]) {
    kt::ViewPatch v{};
    if (patch) v = view_patch_of(e, n);
    if (!e->h_overflow) {
      KT_HIP(e, hipHostMalloc((void**)&e->h_overflow, 64, hipHostMallocDefault));
      memset(e->h_overflow, 0, 64);  // (word 1 is the sequence number settle_ingest compares with)
    }
    if (!e->sw[kSw_INGEST_EVENT_WAIT]) spin_seq = ++e->ingest_seq;
    kt::launch_unfeed_small(e->pods, n, rows_dev, patch ? &v : nullptr, spin_seq ? e->h_overflow + 1 : nullptr, spin_seq, e->own_stream);
    KT_HIP(e, hipGetLastError());
  } else {
    int32_t drc = delta_scan(e, n, rows_dev, 0, -1, e->own_stream);
    if (drc != KT_OK) return drc;
    kt::launch_delete_pods(e->pods, n, rows_dev, e->own_stream);
    if (patch) {  // the rows' meta words are 0 now: their records stop counting
      int32_t prc = patch_views(e, n, rows_dev, 0, e->own_stream);
      if (prc != KT_OK) return prc;
    }
  }
  if (slot_path) {
    KT_HIP(e, hipEventRecord(slot->ev, e->own_stream));
    slot->used = true;
    std::lock_guard<std::mutex> g(e->ingest_mu);
    e->ingest_ev = slot->ev;
    e->ingest_stream = e->own_stream;
    e->ingest_spin_seq = spin_seq;
    e->ingest_pending.store(true, std::memory_order_release);
  } else {
    KT_HIP(e, hipStreamSynchronize(e->own_stream));
  }
  e->last_stream = e->own_stream;
  return KT_OK;
}

static int32_t upsert_throttles_locked(kt_engine* e, const kt_snapshot* b, const int32_t* rows) {
  const int D = e->D;
  if (b->n_thr > 0 && b->D != D) return e->fail(KT_ERR_INVALID_ARGUMENT, "batch D=%d, engine D=%d", b->D, D);
  for (int32_t i = 0; i < b->n_thr; ++i) {
    const int32_t row = rows ? rows[i] : i;
    if (row < 0 || row >= e->cfg.throttle_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", row);
    if (!(b->thr_flags[i] & KT_THR_CLUSTER) && b->thr_ns[i] >= (uint32_t)e->cfg.namespace_capacity)
      return e->fail(KT_ERR_OUT_OF_RANGE, "throttle %d: namespace id %u", i, b->thr_ns[i]);
  }
  if (b->n_thr <= 0) return KT_OK;
  for (int32_t i = 0; i < b->n_thr; ++i) {  // the whole batch is validated before the first row is stored
    HostAmount u, r;
    amount_from_table(b->thr_used, (size_t)i, D, u);
    amount_from_table(b->thr_reserved, (size_t)i, D, r);
    if (!amount_in_bound(u, D) || !amount_in_bound(r, D))
      return e->fail(KT_ERR_OVERFLOW_RISK, "throttle %d: status.used / reserved beyond 2^60", i);
  }
  int32_t rc = sync_status_to_host(e);
  if (rc != KT_OK) return rc;
  for (int32_t i = 0; i < b->n_thr; ++i) {
    HostThrottle h;
    h.flags = b->thr_flags[i];
    h.ns = b->thr_ns[i];
    amount_from_table(b->thr_spec, (size_t)i, D, h.spec);
    amount_from_table(b->thr_calc, (size_t)i, D, h.calc);
    amount_from_table(b->thr_used, (size_t)i, D, h.used);
    amount_from_table(b->thr_reserved, (size_t)i, D, h.reserved);
    h.thrl_flag = b->thr_thrl_flag[i] & ((1u << D) - 1u);
    h.thrl_has = b->thr_thrl_has[i] & ((1u << D) - 1u);
    h.status_fp = b->thr_status_msgs_fp[i];
    h.spec_fp = b->thr_spec_msgs_fp[i];
    for (uint32_t o = b->thr_ovr_off[i]; o < b->thr_ovr_off[i + 1]; ++o) {
      Override ov;
      ov.begin_s = b->ovr_begin_s[o];
      ov.begin_ns = b->ovr_begin_ns[o];
      ov.end_s = b->ovr_end_s[o];
      ov.end_ns = b->ovr_end_ns[o];
      ov.flags = b->ovr_flags[o];
      amount_from_table(b->ovr_thr, (size_t)o, D, ov.thr);
      h.ovr.push_back(ov);
    }
    for (uint32_t g = b->thr_term_off[i]; g < b->thr_term_off[i + 1]; ++g) {
      Term tm;
      tm.flags = b->term_flags[g];
      reqs_from_pool(b->preq, b->term_preq_off[g], b->term_preq_off[g + 1], tm.preq);
      reqs_from_pool(b->nreq, b->term_nreq_off[g], b->term_nreq_off[g + 1], tm.nreq);
      h.terms.push_back(std::move(tm));
    }
    const int32_t row = rows ? rows[i] : i;
    HostThrottle& old = e->thr[(size_t)row];
    if (row < e->thr_rows_hi && same_selector(old, h)) {
      // the compiled program and the index stand; the namespace side of the terms stays cached
      h.adm = std::move(old.adm), h.adm_gen = old.adm_gen, h.adm_ns = old.adm_ns;
      e->spec_dirty = true;
    } else {
      e->program_dirty = true;
    }
    old = std::move(h);
    e->thr_rows_hi = std::max(e->thr_rows_hi, row + 1);
  }
  e->status_host_dirty = true;
  // results of earlier launches describe the old throttle set (and its row count): not fetchable any more
  e->reconcile_ready = e->check_ready = false;
  return KT_OK;
}

int32_t kt_upsert_throttles(kt_engine* e, const kt_snapshot* b, const int32_t* rows) {
  if (!e || !b) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  return upsert_throttles_locked(e, b, rows);
}

int32_t kt_delete_throttles(kt_engine* e, int32_t n, const int32_t* rows) {
  if (!e || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  for (int32_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= e->cfg.throttle_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", rows[i]);
  if (n <= 0) return KT_OK;
  int32_t rc = sync_status_to_host(e);
  if (rc != KT_OK) return rc;
  for (int32_t i = 0; i < n; ++i) e->thr[(size_t)rows[i]] = HostThrottle();
  e->program_dirty = true;
  e->status_host_dirty = true;
  e->reconcile_ready = e->check_ready = false;
  return KT_OK;
}

// ---- single-object forms: a kt_snapshot is assembled HERE (C memory) around the caller's flat arrays
int32_t kt_upsert_namespace(kt_engine* e, int32_t ns_row, int32_t exists, int32_t n_labels, const uint32_t* label_keys,
                            const uint32_t* label_pairs) {
  if (!e || n_labels < 0 || (n_labels > 0 && (!label_keys || !label_pairs))) return KT_ERR_INVALID_ARGUMENT;
  kt_snapshot b{};
  uint8_t valid = exists ? 1 : 0;
  uint32_t off[2] = {0u, (uint32_t)n_labels};
  b.n_ns = 1, b.ns_valid = &valid, b.ns_label_off = off;
  b.ns_label_key = const_cast<uint32_t*>(label_keys), b.ns_label_pair = const_cast<uint32_t*>(label_pairs);
  return kt_upsert_namespaces(e, &b, &ns_row);
}

int32_t kt_upsert_pod(kt_engine* e, int64_t pod_row, uint32_t ns, uint32_t flags, int32_t n_labels, const uint32_t* label_keys,
                      const uint32_t* label_pairs, int32_t n_ctr, const uint8_t* ctr_init, const uint32_t* ctr_present,
                      const int64_t* ctr_req, uint32_t ovh_present, const int64_t* ovh) {
  if (!e || n_labels < 0 || n_ctr < 0 || (n_labels > 0 && (!label_keys || !label_pairs)) ||
      (n_ctr > 0 && (!ctr_init || !ctr_present || !ctr_req)))
    return KT_ERR_INVALID_ARGUMENT;
  kt_snapshot b{};
  int64_t zero_ovh[KT_MAX_DIMS] = {0};
  uint32_t loff[2] = {0u, (uint32_t)n_labels}, coff[2] = {0u, (uint32_t)n_ctr};
  b.D = e->D, b.L = e->L, b.n_pods = 1;
  b.pod_ns = &ns, b.pod_flags = &flags, b.pod_label_off = loff;
  b.pod_label_key = const_cast<uint32_t*>(label_keys), b.pod_label_pair = const_cast<uint32_t*>(label_pairs);
  b.pod_ctr_off = coff, b.ctr_init = const_cast<uint8_t*>(ctr_init), b.ctr_present = const_cast<uint32_t*>(ctr_present);
  b.ctr_req = const_cast<int64_t*>(ctr_req);
  if (!ovh) ovh_present &= ~0x80000000u;
  b.pod_ovh_present = &ovh_present, b.pod_ovh = ovh ? const_cast<int64_t*>(ovh) : zero_ovh;
  return kt_upsert_pods(e, &b, &pod_row);
}

int32_t kt_upsert_throttle(kt_engine* e, int32_t thr_row, uint32_t flags, uint32_t ns, const int64_t* amt_v,
                           const uint32_t* amt_present, const int64_t* amt_count, const uint8_t* amt_has_count,
                           uint32_t thrl_flag, uint32_t thrl_has, uint64_t status_msgs_fp, uint64_t spec_msgs_fp, int32_t n_ovr,
                           const int64_t* ovr_begin_s, const int32_t* ovr_begin_ns, const int64_t* ovr_end_s,
                           const int32_t* ovr_end_ns, const uint8_t* ovr_flags, const int64_t* ovr_v, const uint32_t* ovr_present,
                           const int64_t* ovr_count, const uint8_t* ovr_has_count, int32_t n_terms, const uint8_t* term_flags,
                           const uint32_t* term_preq_off, const uint32_t* term_nreq_off, uint32_t n_preq, const uint8_t* preq_op,
                           const uint32_t* preq_key, const uint32_t* preq_val_off, const uint32_t* preq_val, uint32_t n_nreq,
                           const uint8_t* nreq_op, const uint32_t* nreq_key, const uint32_t* nreq_val_off, const uint32_t* nreq_val) {
  if (!e || !amt_v || !amt_present || !amt_count || !amt_has_count || n_ovr < 0 || n_terms < 0 ||
      (n_ovr > 0 && (!ovr_begin_s || !ovr_begin_ns || !ovr_end_s || !ovr_end_ns || !ovr_flags || !ovr_v || !ovr_present || !ovr_count ||
                     !ovr_has_count)) ||
      (n_terms > 0 && (!term_flags || !term_preq_off || !term_nreq_off)) ||
      (n_preq > 0 && (!preq_op || !preq_key || !preq_val_off)) || (n_nreq > 0 && (!nreq_op || !nreq_key || !nreq_val_off)))
    return KT_ERR_INVALID_ARGUMENT;
  const int D = e->D;
  kt_snapshot b{};
  b.D = D, b.L = e->L, b.n_thr = 1;
  b.thr_flags = &flags, b.thr_ns = &ns;
  kt_amounts* rows[4] = {&b.thr_spec, &b.thr_calc, &b.thr_used, &b.thr_reserved};
  for (int k = 0; k < 4; ++k) {
    rows[k]->v = const_cast<int64_t*>(amt_v) + (size_t)k * D;
    rows[k]->present = const_cast<uint32_t*>(amt_present) + k;
    rows[k]->count = const_cast<int64_t*>(amt_count) + k;
    rows[k]->has_count = const_cast<uint8_t*>(amt_has_count) + k;
  }
  b.thr_thrl_flag = &thrl_flag, b.thr_thrl_has = &thrl_has, b.thr_status_msgs_fp = &status_msgs_fp, b.thr_spec_msgs_fp = &spec_msgs_fp;
  uint32_t ooff[2] = {0u, (uint32_t)n_ovr}, toff[2] = {0u, (uint32_t)n_terms};
  b.thr_ovr_off = ooff;
  b.ovr_begin_s = const_cast<int64_t*>(ovr_begin_s), b.ovr_begin_ns = const_cast<int32_t*>(ovr_begin_ns);
  b.ovr_end_s = const_cast<int64_t*>(ovr_end_s), b.ovr_end_ns = const_cast<int32_t*>(ovr_end_ns);
  b.ovr_flags = const_cast<uint8_t*>(ovr_flags);
  b.ovr_thr = kt_amounts{const_cast<int64_t*>(ovr_v), const_cast<uint32_t*>(ovr_present), const_cast<int64_t*>(ovr_count),
                         const_cast<uint8_t*>(ovr_has_count)};
  b.thr_term_off = toff;
  uint32_t zero2[2] = {0u, 0u};
  b.term_flags = const_cast<uint8_t*>(term_flags);
  b.term_preq_off = n_terms ? const_cast<uint32_t*>(term_preq_off) : zero2;
  b.term_nreq_off = n_terms ? const_cast<uint32_t*>(term_nreq_off) : zero2;
  uint32_t zero1[1] = {0u};
  b.preq = kt_reqs{n_preq, const_cast<uint8_t*>(preq_op), const_cast<uint32_t*>(preq_key),
                   n_preq ? const_cast<uint32_t*>(preq_val_off) : zero1, const_cast<uint32_t*>(preq_val)};
  b.nreq = kt_reqs{n_nreq, const_cast<uint8_t*>(nreq_op), const_cast<uint32_t*>(nreq_key),
                   n_nreq ? const_cast<uint32_t*>(nreq_val_off) : zero1, const_cast<uint32_t*>(nreq_val)};
  if (n_terms > 0 && (term_preq_off[n_terms] > n_preq || term_nreq_off[n_terms] > n_nreq))
    return e->fail(KT_ERR_OUT_OF_RANGE, "selector terms reference %u / %u requirements, pools hold %u / %u", term_preq_off[n_terms],
                   term_nreq_off[n_terms], n_preq, n_nreq);
  // 37 positional arguments: one slice in the wrong position is a silent mis-feed unless the shapes are held against each
  // other here — offsets start at 0 and never decrease, every operator is one of the four, masks name existing dimensions
  auto bad = [&](const char* what, long long i, long long v) {
    return e->fail(KT_ERR_INVALID_ARGUMENT, "kt_upsert_throttle(row %d): %s[%lld] = %lld does not fit the other arguments", thr_row, what, i, v);
  };
  const uint32_t dmask = D >= 32 ? ~0u : (1u << D) - 1u;
  for (int k = 0; k < 4; ++k)
    if (amt_present[k] & ~dmask) return bad("amt_present", k, amt_present[k]);
  if ((thrl_has | thrl_flag) & ~dmask) return bad("thrl_has | thrl_flag", 0, thrl_has | thrl_flag);
  for (int32_t o = 0; o < n_ovr; ++o)
    if (ovr_present[o] & ~dmask) return bad("ovr_present", o, ovr_present[o]);
  if (n_terms > 0 && (term_preq_off[0] != 0u || term_nreq_off[0] != 0u)) return bad("term_preq_off / term_nreq_off", 0, term_preq_off[0] | term_nreq_off[0]);
  for (int32_t t = 0; t < n_terms; ++t) {
    if (term_preq_off[t + 1] < term_preq_off[t]) return bad("term_preq_off", t + 1, term_preq_off[t + 1]);
    if (term_nreq_off[t + 1] < term_nreq_off[t]) return bad("term_nreq_off", t + 1, term_nreq_off[t + 1]);
    if (term_flags[t] & ~(KT_TERM_POD_SEL_INVALID | KT_TERM_NS_SEL_INVALID)) return bad("term_flags", t, term_flags[t]);
  }
  struct Pool { const char* name; uint32_t n; const uint8_t* op; const uint32_t* val_off; const uint32_t* val; };
  const Pool pools[2] = {{"preq", n_preq, preq_op, preq_val_off, preq_val}, {"nreq", n_nreq, nreq_op, nreq_val_off, nreq_val}};
  for (const Pool& pl : pools) {
    if (pl.n && pl.val_off[0] != 0u) return bad(pl.name, 0, pl.val_off[0]);
    for (uint32_t r = 0; r < pl.n; ++r) {
      if (pl.op[r] > KT_OP_DOES_NOT_EXIST) return bad(pl.name, r, pl.op[r]);
      if (pl.val_off[r + 1] < pl.val_off[r]) return bad(pl.name, r + 1, pl.val_off[r + 1]);
    }
    if (pl.n && pl.val_off[pl.n] > 0u && !pl.val) return bad(pl.name, pl.n, pl.val_off[pl.n]);
  }
  return kt_upsert_throttles(e, &b, &thr_row);
}


int32_t kt_load_snapshot(kt_engine* e, const kt_snapshot* s) {
  if (!e || !s) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (s->n_ns > e->cfg.namespace_capacity || s->n_pods > e->cfg.pod_capacity || s->n_thr > e->cfg.throttle_capacity)
    return e->fail(KT_ERR_OUT_OF_RANGE, "snapshot larger than the configured capacity");
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  // clear
  KT_HIP(e, hipMemsetAsync(e->pods.flags, 0, (size_t)e->cfg.pod_capacity * 4, e->own_stream));
  KT_HIP(e, hipMemsetAsync(e->pods.meta, 0, (size_t)e->cfg.pod_capacity * 8, e->own_stream));
  KT_HIP(e, hipStreamSynchronize(e->own_stream));
  e->countable_valid = false;
  e->req_sums_valid = true;
  for (auto& b : e->req_sum_bound) b = 0;
  e->order_all_valid = false;
  e->pod_rows_hi = 0;
  e->pod_ns_hi = 0;
  e->neg_seen = false;
  for (auto& m : e->max_abs) m = 0;
  for (auto& m : e->or_abs) m = 0;
  for (auto& n : e->ns) n = HostNamespace();
  ++e->ns_gen;
  for (auto& t : e->thr) t = HostThrottle();
  e->ns_rows_hi = 0;
  e->thr_rows_hi = 0;
  e->status_dev_newer = false;
  e->reserved_dev_newer = false;
  e->program_dirty = true;
  e->status_host_dirty = true;
  e->reconcile_ready = e->check_ready = false;
  for (int32_t i = 0; i < s->n_ns; ++i) {
    HostNamespace& n = e->ns[(size_t)i];
    n.valid = s->ns_valid ? s->ns_valid[i] != 0 : true;
    for (uint32_t k = s->ns_label_off[i]; k < s->ns_label_off[i + 1]; ++k)
      n.labels.emplace_back(s->ns_label_key[k], s->ns_label_pair[k]);
  }
  e->ns_rows_hi = s->n_ns;
  int32_t rc = upsert_throttles_locked(e, s, nullptr);
  if (rc != KT_OK) return rc;
  return upsert_pods_locked(e, s, nullptr);
}

int32_t kt_set_reserved(kt_engine* e, int32_t n, const int32_t* rows, const kt_amounts* reserved) {
  if (!e || !reserved || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  for (int32_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= e->thr_rows_hi) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", rows[i]);
  int32_t rc = sync_status_to_host(e);
  if (rc != KT_OK) return rc;
  for (int32_t i = 0; i < n; ++i) {
    HostAmount a;
    amount_from_table(*reserved, (size_t)i, e->D, a);
    if (!amount_in_bound(a, e->D)) return e->fail(KT_ERR_OVERFLOW_RISK, "reserved amount beyond 2^60");
    e->thr[(size_t)rows[i]].reserved = a;
  }
  e->status_host_dirty = true;
  return KT_OK;
}

int32_t kt_set_status(kt_engine* e, int32_t n, const int32_t* rows, const kt_status* st) {
  if (!e || !st || (n > 0 && !rows)) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  for (int32_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= e->thr_rows_hi) return e->fail(KT_ERR_OUT_OF_RANGE, "throttle row %d", rows[i]);
  int32_t rc = sync_status_to_host(e);
  if (rc != KT_OK) return rc;
  const uint32_t dm = (1u << e->D) - 1u;
  for (int32_t i = 0; i < n; ++i) {
    HostThrottle& h = e->thr[(size_t)rows[i]];
    HostAmount u, c;
    amount_from_table(st->used, (size_t)i, e->D, u);
    amount_from_table(st->calc, (size_t)i, e->D, c);
    if (!amount_in_bound(u, e->D)) return e->fail(KT_ERR_OVERFLOW_RISK, "status.used beyond 2^60");
    h.used = u;
    h.calc = c;
    h.thrl_flag = st->thrl_flag ? st->thrl_flag[i] & dm : 0;
    h.thrl_has = st->thrl_has ? st->thrl_has[i] & dm : 0;
    h.flags &= ~(uint32_t)(KT_THR_CALC_AT_NONZERO | KT_THR_THROTTLED_POD);
    if (st->calc_at_nonzero && st->calc_at_nonzero[i]) h.flags |= KT_THR_CALC_AT_NONZERO;
    if (st->thrl_pod && st->thrl_pod[i]) h.flags |= KT_THR_THROTTLED_POD;
    h.status_fp = st->msgs_fp ? st->msgs_fp[i] : 0;
  }
  e->status_host_dirty = true;
  return KT_OK;
}

