// kt_engine.cpp — host side of libkt_engine.so: the C-ABI of include/kt_engine.h over the HIP kernels.
//
// Responsibilities: own every device allocation (pod row tables, throttle tables, selector program,
// index, result buffers), validate and stage caller batches (the caller's memory is never retained),
// compile throttles + namespaces into the device selector program, launch kernels on the caller's
// stream, time them with HIP events.  No compute happens here: without a gfx950 device
// kt_engine_create fails (KT_ERR_NO_DEVICE) — there is no CPU fallback.
#include "kt_engine_impl.h"

thread_local std::string g_create_error;

static bool getenv_flag(const char* name) {
  const char* v = getenv(name);
  return v && *v && *v != '0';
}
static void load_env_switches(kt_engine* e) {
  for (int k = 0; k < kSwCount; ++k) e->sw[k] = getenv_flag(kEnvSwitchName[k]);
}

// CheckRecs about to be rewritten IN PLACE: no few-pod check may start on them (recs_valid = false under recs_mu) and
// the one in flight, if any, has to finish first (it holds small_mu from launch to completion)
void recs_invalidate_and_drain(kt_engine* e) {
  {
    std::lock_guard<std::mutex> g(e->recs_mu);
    e->recs_valid = false;
  }
  if (e->few_ready) std::lock_guard<std::mutex> drain(e->small_mu);
}

hipStream_t pick_stream(kt_engine* e, void* s) {
  hipStream_t st = s ? (hipStream_t)s : e->own_stream;
  order_behind_ingest(e, st);
  return st;
}


int32_t upload_amounts(kt_engine* e, AmountDev& d, const AmountHostFlat& h, size_t n, int D, hipStream_t s) {
  KT_HIP(e, d.reserve(n + 1, D));
  if (n) {
    KT_HIP(e, hipMemcpyAsync(d.v.p, h.v.data(), n * D * 8, hipMemcpyHostToDevice, s));
    KT_HIP(e, hipMemcpyAsync(d.present.p, h.present.data(), n * 4, hipMemcpyHostToDevice, s));
    KT_HIP(e, hipMemcpyAsync(d.count.p, h.count.data(), n * 8, hipMemcpyHostToDevice, s));
    KT_HIP(e, hipMemcpyAsync(d.has_count.p, h.has_count.data(), n, hipMemcpyHostToDevice, s));
  }
  return KT_OK;
}
int32_t download_amounts(kt_engine* e, const AmountDev& d, AmountHostFlat& h, size_t n, int D, hipStream_t s) {
  h.resize(n, D);
  if (n) {
    KT_HIP(e, hipMemcpyAsync(h.v.data(), d.v.p, n * D * 8, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(h.present.data(), d.present.p, n * 4, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(h.count.data(), d.count.p, n * 8, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(h.has_count.data(), d.has_count.p, n, hipMemcpyDeviceToHost, s));
  }
  return KT_OK;
}

// Pull the device-resident status back into the host mirrors (after a reconcile with APPLY).
int32_t sync_status_to_host(kt_engine* e) {
  if (!e->status_dev_newer && !e->reserved_dev_newer) return KT_OK;
  const size_t T = (size_t)e->thr_rows_hi;
  const int D = e->D;
  hipStream_t s = e->own_stream;
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  if (e->reserved_dev_newer) {  // kt_admit_launch(KT_ADMIT_COMMIT) advanced the reserved amounts on the device
    AmountHostFlat res;
    int32_t rc0;
    if ((rc0 = download_amounts(e, e->d_reserved, res, T, D, s)) != KT_OK) return rc0;
    KT_HIP(e, hipStreamSynchronize(s));
    for (size_t t = 0; t < T; ++t)
      if (e->thr[t].flags & KT_THR_VALID) res.get(t, D, e->thr[t].reserved);
    e->reserved_dev_newer = false;
  }
  if (!e->status_dev_newer) return KT_OK;
  AmountHostFlat used, calc;
  std::vector<uint32_t> flags(T), tf(T), th(T);
  std::vector<uint64_t> fp(T);
  std::vector<int64_t> used_hi(T * D + 1, 0);
  int32_t rc;
  if (T && e->d_used_hi.p) KT_HIP(e, hipMemcpyAsync(used_hi.data(), e->d_used_hi.p, T * D * 8, hipMemcpyDeviceToHost, s));
  if ((rc = download_amounts(e, e->d_used, used, T, D, s)) != KT_OK) return rc;
  if ((rc = download_amounts(e, e->d_calc, calc, T, D, s)) != KT_OK) return rc;
  if (T) {
    KT_HIP(e, hipMemcpyAsync(flags.data(), e->d_thr_flags.p, T * 4, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(tf.data(), e->d_thrl_flag.p, T * 4, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(th.data(), e->d_thrl_has.p, T * 4, hipMemcpyDeviceToHost, s));
    KT_HIP(e, hipMemcpyAsync(fp.data(), e->d_status_fp.p, T * 8, hipMemcpyDeviceToHost, s));
  }
  KT_HIP(e, hipStreamSynchronize(s));
  for (size_t t = 0; t < T; ++t) {
    HostThrottle& h = e->thr[t];
    if (!(h.flags & KT_THR_VALID)) continue;
    used.get(t, D, h.used);
    for (int d = 0; d < D; ++d) h.used.v_hi[d] = e->d_used_hi.p ? used_hi[t * D + d] : (h.used.v[d] < 0 ? -1 : 0);
    calc.get(t, D, h.calc);
    h.flags = flags[t];
    h.thrl_flag = tf[t];
    h.thrl_has = th[t];
    h.status_fp = fp[t];
  }
  e->status_dev_newer = false;
  return KT_OK;
}

// Flatten status + reserved rows and push them to the device.
int32_t upload_status(kt_engine* e, hipStream_t s) {
  const size_t T = (size_t)e->thr_rows_hi;
  const int D = e->D;
  AmountHostFlat calc, used, res;
  calc.resize(T, D);
  used.resize(T, D);
  res.resize(T, D);
  std::vector<uint32_t> flags(T), tf(T), th(T);
  std::vector<uint64_t> fp(T);
  std::vector<int64_t> used_hi(T * D + 1, 0);
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    calc.set(t, D, h.calc);
    used.set(t, D, h.used);
    for (int d = 0; d < D; ++d) used_hi[t * D + d] = ((h.used.present >> d) & 1u) ? h.used.v_hi[d] : 0;
    res.set(t, D, h.reserved);
    flags[t] = h.flags;
    tf[t] = h.thrl_flag;
    th[t] = h.thrl_has;
    fp[t] = h.status_fp;
  }
  int32_t rc;
  if ((rc = upload_amounts(e, e->d_calc, calc, T, D, s)) != KT_OK) return rc;
  if ((rc = upload_amounts(e, e->d_used, used, T, D, s)) != KT_OK) return rc;
  if ((rc = upload(e, e->d_used_hi, used_hi, s)) != KT_OK) return rc;
  if ((rc = upload_amounts(e, e->d_reserved, res, T, D, s)) != KT_OK) return rc;
  if ((rc = upload(e, e->d_thr_flags, flags, s)) != KT_OK) return rc;
  if ((rc = upload(e, e->d_thrl_flag, tf, s)) != KT_OK) return rc;
  if ((rc = upload(e, e->d_thrl_has, th, s)) != KT_OK) return rc;
  if ((rc = upload(e, e->d_status_fp, fp, s)) != KT_OK) return rc;
  KT_HIP(e, hipStreamSynchronize(s));  // host vectors go out of scope
  e->status_host_dirty = false;
  e->recs_valid = false;
  return KT_OK;
}

// Compile throttles + namespaces into the device selector program, spec tables and index.
// spec.threshold, the temporary threshold overrides and the fingerprint of the spec's messages of every throttle row:
// what kt_finalize reads beside the status.  Part of a compile; on its own after Throttle events that left every
// selector as it was (spec_dirty).
int32_t upload_spec_tables(kt_engine* e, hipStream_t s) {
  const int D = e->D;
  const size_t T = (size_t)e->thr_rows_hi;
  std::vector<uint32_t> ovr_off(T + 1, 0);
  std::vector<int64_t> ob_s, oe_s;
  std::vector<int32_t> ob_ns, oe_ns;
  std::vector<uint8_t> o_flags;
  std::vector<uint64_t> spec_fp(T);
  AmountHostFlat spec, ovr_thr;
  spec.resize(T, D);
  size_t n_ovr = 0;
  for (size_t t = 0; t < T; ++t) n_ovr += e->thr[t].ovr.size();
  ovr_thr.resize(n_ovr, D);
  size_t o = 0;
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    spec.set(t, D, h.spec);
    spec_fp[t] = h.spec_fp;
    for (const Override& ov : h.ovr) {
      ob_s.push_back(ov.begin_s);
      ob_ns.push_back(ov.begin_ns);
      oe_s.push_back(ov.end_s);
      oe_ns.push_back(ov.end_ns);
      o_flags.push_back(ov.flags);
      ovr_thr.set(o++, D, ov.thr);
    }
    ovr_off[t + 1] = (uint32_t)o;
  }
  int32_t rc;
#define UP(dev, host) if ((rc = upload(e, e->dev, host, s)) != KT_OK) return rc
  UP(d_ovr_off, ovr_off);
  UP(d_ovr_begin_s, ob_s);
  UP(d_ovr_begin_ns, ob_ns);
  UP(d_ovr_end_s, oe_s);
  UP(d_ovr_end_ns, oe_ns);
  UP(d_ovr_flags, o_flags);
  UP(d_spec_fp, spec_fp);
#undef UP
  if ((rc = upload_amounts(e, e->d_spec, spec, T, D, s)) != KT_OK) return rc;
  if ((rc = upload_amounts(e, e->d_ovr_thr, ovr_thr, n_ovr, D, s)) != KT_OK) return rc;
  KT_HIP(e, hipStreamSynchronize(s));  // host vectors go out of scope
  e->spec_dirty = false;
  return KT_OK;
}


int32_t ensure_ready(kt_engine* e, hipStream_t s) {
  int32_t rc;
  if (e->program_dirty || e->spec_dirty || e->status_host_dirty) {
    // uploads reallocate/overwrite device tables that an in-flight kernel of the last stream may read
    if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  }
  if (e->program_dirty) {
    if ((rc = sync_status_to_host(e)) != KT_OK) return rc;
    if ((rc = compile_program(e, e->own_stream)) != KT_OK) return rc;  // (takes the spec tables along)
    e->status_host_dirty = true;
  } else if (e->spec_dirty) {
    if ((rc = upload_spec_tables(e, e->own_stream)) != KT_OK) return rc;
  }
  if (e->status_host_dirty) {
    if ((rc = upload_status(e, e->own_stream)) != KT_OK) return rc;
  }
  e->tt.flags = e->d_thr_flags.p;
  e->tt.spec = e->d_spec.tab();
  e->tt.calc = e->d_calc.tab();
  e->tt.used = e->d_used.tab();
  e->tt.used_hi = e->d_used_hi.p;
  e->tt.reserved = e->d_reserved.tab();
  e->tt.thrl_flag = e->d_thrl_flag.p;
  e->tt.thrl_has = e->d_thrl_has.p;
  e->tt.status_msgs_fp = e->d_status_fp.p;
  e->tt.spec_msgs_fp = e->d_spec_fp.p;
  e->tt.ovr_off = e->d_ovr_off.p;
  e->tt.ovr_begin_s = e->d_ovr_begin_s.p;
  e->tt.ovr_begin_ns = e->d_ovr_begin_ns.p;
  e->tt.ovr_end_s = e->d_ovr_end_s.p;
  e->tt.ovr_end_ns = e->d_ovr_end_ns.p;
  e->tt.ovr_flags = e->d_ovr_flags.p;
  e->tt.ovr_thr = e->d_ovr_thr.tab();
  (void)s;
  return KT_OK;
}

void amount_from_table(const kt_amounts& a, size_t i, int D, HostAmount& h) {
  h.present = a.present ? a.present[i] & ((1u << D) - 1u) : 0;
  for (int d = 0; d < D; ++d) h.v[d] = ((h.present >> d) & 1u) ? a.v[i * D + d] : 0;
  for (int d = 0; d < D; ++d) h.v_hi[d] = h.v[d] < 0 ? -1 : 0;
  h.has_count = a.has_count ? (a.has_count[i] != 0) : 0;
  h.count = h.has_count ? a.count[i] : 0;
}

void amount_to_table(const HostAmount& h, const kt_amounts& a, size_t i, int D) {
  if (a.present) a.present[i] = h.present;
  for (int d = 0; d < D; ++d) a.v[i * D + d] = ((h.present >> d) & 1u) ? h.v[d] : 0;
  if (a.has_count) a.has_count[i] = h.has_count;
  if (a.count) a.count[i] = h.has_count ? h.count : 0;
}


// upper bound of every pod's effective request per dimension, for kRecTight (kt_device.h)
kt::ReqBound req_bound(const kt_engine* e) {
  kt::ReqBound b;
  for (int d = 0; d < 16; ++d)
    b.v[d] = d < e->D ? (e->max_abs[d] > (unsigned __int128)INT64_MAX ? INT64_MAX : (int64_t)e->max_abs[d]) : 0;
  return b;
}

bool amount_in_bound(const HostAmount& a, int D) {
  for (int d = 0; d < D; ++d)
    if (((a.present >> d) & 1u) && uabs(a.v[d]) > kSumBound) return false;
  return uabs(a.count) <= kSumBound;
}

void reqs_from_pool(const kt_reqs& pool, uint32_t b, uint32_t e_, std::vector<Req>& out) {
  out.clear();
  for (uint32_t r = b; r < e_; ++r) {
    Req q;
    q.op = pool.op[r];
    q.key = pool.key[r];
    q.vals.assign(pool.val + pool.val_off[r], pool.val + pool.val_off[r + 1]);
    out.push_back(std::move(q));
  }
}


// ===================================================================================================
// C-ABI (the exports are declared extern "C" by include/kt_engine.h)
// ===================================================================================================

#ifndef KT_SRC_HASH
#define KT_SRC_HASH "unknown"
#endif
const char* kt_version(void) { return "kt-engine 0.2 (gfx950, HIP) src=" KT_SRC_HASH; }

const char* kt_last_error(kt_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int32_t kt_engine_create(const kt_config* cfg, kt_engine** out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return KT_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (cfg->n_dims < 1 || cfg->n_dims > KT_MAX_DIMS || cfg->max_labels < 1 || cfg->max_labels > KT_MAX_LABELS ||
      cfg->pod_capacity < 1 || cfg->pod_capacity > (1ll << 31) || cfg->throttle_capacity < 1 || cfg->throttle_capacity >= (1 << 20) ||
      cfg->namespace_capacity < 1) {
    g_create_error = "invalid kt_config";
    return KT_ERR_INVALID_ARGUMENT;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible: the engine has no CPU fallback";
    return KT_ERR_NO_DEVICE;
  }
  int dev = cfg->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  }
  if (dev >= ndev) {
    g_create_error = "device ordinal out of range";
    return KT_ERR_INVALID_ARGUMENT;
  }
  kt_engine* e = new kt_engine();
  e->cfg = *cfg;
  load_env_switches(e);
  e->device = dev;
  e->D = cfg->n_dims;
  e->L = cfg->max_labels;
  e->incremental = (cfg->kernel_variant & KT_VARIANT_INCREMENTAL) != 0;
  e->cfg.kernel_variant &= 0xFF;
  if (e->incremental && e->cfg.kernel_variant != 0) {
    g_create_error = "KT_VARIANT_INCREMENTAL needs the indexed kernels (kernel_variant 0)";
    delete e;
    return KT_ERR_INVALID_ARGUMENT;
  }
  hipError_t r = hipSetDevice(dev);
  if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
  const size_t cap = (size_t)cfg->pod_capacity;
  e->pods.cap = cfg->pod_capacity;
  e->pods.D = e->D;
  e->pods.L = e->L;
  e->pods.DS = kt::req_stride(e->D);
  e->pods.LS = kt::label_stride(e->L);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.ns, cap * 4);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.flags, cap * 4);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.req, cap * 8 * e->pods.DS);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.lpair, cap * 4 * e->pods.LS);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.lkey, cap * 4 * e->pods.LS);
  if (r == hipSuccess) r = kt::kt_alloc_device((void**)&e->pods.meta, cap * 8);
  e->pods.LA = 8;
  if (r == hipSuccess) r = hipMemsetAsync(e->pods.flags, 0, cap * 4, e->own_stream);
  if (r == hipSuccess) r = hipMemsetAsync(e->pods.meta, 0, cap * 8, e->own_stream);
  // rows that were never upserted are read by kt_translate_pods: empty label slots
  if (r == hipSuccess) r = hipMemsetAsync(e->pods.lpair, 0, cap * 4 * e->pods.LS, e->own_stream);
  if (r == hipSuccess) r = hipMemsetAsync(e->pods.lkey, 0, cap * 4 * e->pods.LS, e->own_stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->own_stream);
  if (r != hipSuccess) {
    g_create_error = std::string("device setup failed: ") + hipGetErrorString(r);
    kt_engine_destroy(e);
    return KT_ERR_DEVICE;
  }
  e->ns.resize((size_t)cfg->namespace_capacity);
  e->thr.resize((size_t)cfg->throttle_capacity);
  *out = e;
  return KT_OK;
}

int32_t kt_engine_destroy(kt_engine* e) {
  if (!e) return KT_OK;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  if (e->comm) (void)rccl()->CommDestroy(e->comm);
  if (e->pods.ns) (void)hipFree(e->pods.ns);
  if (e->pods.flags) (void)hipFree(e->pods.flags);
  if (e->pods.req) (void)hipFree(e->pods.req);
  if (e->pods.lpair) (void)hipFree(e->pods.lpair);
  if (e->pods.lkey) (void)hipFree(e->pods.lkey);
  if (e->pods.meta) (void)hipFree(e->pods.meta);
  e->d_latom.release();
  e->d_overflow.release();
  e->d_countable.release();
  e->d_order_all.release();
  e->d_vc_meta.release(); e->d_va_meta.release(); e->d_carry.release();
  e->d_vc_latom.release(); e->d_va_latom.release(); e->d_vc_req.release(); e->d_vc_pk.release();
  e->d_pos_c.release(); e->d_pos_a.release(); e->d_view_dirty.release(); e->d_n_all.release();
  e->d_ns_cursor.release(); e->d_range_a.release(); e->d_range_c.release();
  e->d_slab_tag.release();
  e->d_row_mask.release();
  e->d_req_sums.release();
  e->d_n_countable.release();
  e->d_ticket.release();
  if (e->h_small) (void)hipHostFree(e->h_small);
  if (e->h_few) (void)hipHostFree(e->h_few);
  e->d_few_acc.release();
  e->d_few_ticket.release();
  for (auto& ev : e->recs_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (e->small_stream) (void)hipStreamDestroy(e->small_stream);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  for (auto& sl : e->ev_slots) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.h) (void)hipHostFree(sl.h);
  }
  if (e->h_overflow) (void)hipHostFree(e->h_overflow);
  DevBuf<uint32_t>* u32s[] = {&e->d_thr_term_off, &e->d_term_thr, &e->d_term_req_off, &e->d_req_key, &e->d_req_val_off,
                              &e->d_req_val, &e->d_ns_term_ok, &e->d_thr_flags, &e->d_thrl_flag, &e->d_thrl_has,
                              &e->d_ovr_off, &e->d_out_thrl_flag, &e->d_out_thrl_has};
  for (auto* b : u32s) b->release();
  DevBuf<uint8_t>* u8s[] = {&e->d_term_flags, &e->d_req_op, &e->d_ns_valid, &e->d_ovr_flags, &e->d_out_calc_updated,
                            &e->d_out_thrl_pod, &e->d_out_error, &e->d_recs2[0], &e->d_recs2[1], &e->d_wvimg[0], &e->d_wvimg[1], &e->d_status, &e->d_stage, &e->d_ev_stage, &e->d_slab, &e->d_admit};
  for (auto* b : u8s) b->release();
  e->d_status_fp.release(); e->d_spec_fp.release(); e->d_summary.release(); e->d_rows.release();
  e->d_used_hi.release(); e->d_out_used_hi.release();
  e->d_ovr_begin_s.release(); e->d_ovr_end_s.release(); e->d_ovr_begin_ns.release(); e->d_ovr_end_ns.release();
  e->d_partial.release();
  e->d_out_next_s.release();
  e->d_agg.release();
  e->d_out_next_ns.release();
  e->d_sp.release();
  AmountDev* ams[] = {&e->d_spec, &e->d_calc, &e->d_used, &e->d_reserved, &e->d_ovr_thr, &e->d_out_used, &e->d_out_calc};
  for (auto* a : ams) a->release();
  kt::release_index(e->dindex);
  for (auto& f : e->fam)
    for (auto& pr : f.pool) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return KT_OK;
}

// ---------------------------------------------------------------------------------------------------
// state feed
// ---------------------------------------------------------------------------------------------------

int32_t kt_throttle_rows(kt_engine* e, int32_t* out_rows) {
  if (!e || !out_rows) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  *out_rows = e->thr_rows_hi;
  return KT_OK;
}

int32_t kt_check_device_summary(kt_engine* e, void** device_ptr) {
  if (!e || !device_ptr) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  if (!e->check_ready) return e->fail(KT_ERR_NOT_READY, "no check launched yet");
  *device_ptr = e->d_summary.p;
  return KT_OK;
}

int32_t kt_fetch_pod_requests(kt_engine* e, int64_t n, const int64_t* pod_rows, int64_t* out_v, uint32_t* out_present) {
  if (!e || n < 0 || !out_v || !out_present) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (n == 0) return KT_OK;
  if (!pod_rows && n > e->cfg.pod_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "n > pod_capacity");
  if (pod_rows)
    for (int64_t i = 0; i < n; ++i)
      if (pod_rows[i] < 0 || pod_rows[i] >= e->cfg.pod_capacity) return e->fail(KT_ERR_OUT_OF_RANGE, "pod row");
  hipStream_t s = e->own_stream;
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  const int D = e->D;
  size_t bytes = (size_t)n * 8 * D + (size_t)n * 4 + (pod_rows ? (size_t)n * 8 : 0) + 64;
  KT_HIP(e, e->d_stage.reserve(bytes));
  int64_t* dv = (int64_t*)e->d_stage.p;
  int64_t* drows = dv + (size_t)n * D;
  uint32_t* dp = (uint32_t*)(drows + (pod_rows ? n : 0));
  if (pod_rows) KT_HIP(e, hipMemcpyAsync(drows, pod_rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
  kt::launch_gather_pod_requests(e->pods, n, pod_rows ? drows : nullptr, dv, dp, s);
  KT_HIP(e, hipMemcpyAsync(out_v, dv, (size_t)n * 8 * D, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipMemcpyAsync(out_present, dp, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));
  return KT_OK;
}

// ---------------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------------
int32_t kt_timing_enable(kt_engine* e, int32_t on) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  e->timing = on != 0;
  return KT_OK;
}

int32_t kt_timing_reset(kt_engine* e) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  for (auto& f : e->fam) f.used = 0;
  return KT_OK;
}

int32_t kt_timing_read(kt_engine* e, int32_t kernel, double* total_ms, int64_t* launches) {
  if (!e || kernel < 0 || kernel >= KT_KERNEL_COUNT || !total_ms || !launches) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  if (e->last_stream) KT_HIP(e, hipStreamSynchronize(e->last_stream));
  double tot = 0;
  TimingFamily& f = e->fam[kernel];
  for (size_t i = 0; i < f.used; ++i) {
    float ms = 0;
    KT_HIP(e, hipEventElapsedTime(&ms, f.pool[i].first, f.pool[i].second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)f.used;
  return KT_OK;
}

int32_t kt_synchronize(kt_engine* e, void* stream) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  LaunchLock lk(e);
  KT_HIP(e, hipSetDevice(e->device));
  KT_HIP(e, hipStreamSynchronize(pick_stream(e, stream)));
  return KT_OK;
}

// ---- pages: more resource names than KT_MAX_DIMS (include/kt_engine.h)
int32_t kt_debug_reload_env(kt_engine* e) {
  if (!e) return KT_ERR_INVALID_ARGUMENT;
  StateLock lk(e);  // nobody else inside: the switches are plain fields
  load_env_switches(e);
  return KT_OK;
}

int64_t kt_counter(kt_engine* e, int32_t which) {
  if (!e) return -1;
  switch (which) {
    case KT_COUNTER_FEW_CHECKS: return e->few_served.load(std::memory_order_relaxed);
    case KT_COUNTER_COMPILES: return e->n_compiles.load(std::memory_order_relaxed);
    // (the index figures are copied into atomics at the end of every compile: a metrics thread may ask during a recompile)
    case KT_COUNTER_INDEX_CHUNKS: return e->ctr_index_chunks.load(std::memory_order_relaxed);
    case KT_COUNTER_INDEX_WORDS: return e->ctr_index_words.load(std::memory_order_relaxed);
    case KT_COUNTER_NS_ROWS: return e->ctr_ns_rows.load(std::memory_order_relaxed);
    case KT_COUNTER_NS_WORD_VISITS: return e->ctr_ns_word_visits.load(std::memory_order_relaxed);
    case KT_COUNTER_NS_CHUNK_VISITS: return e->ctr_ns_chunk_visits.load(std::memory_order_relaxed);
    case KT_COUNTER_INDEX_IMAGE_WORDS: return e->ctr_index_image_words.load(std::memory_order_relaxed);
    case KT_COUNTER_SLOW_THROTTLES: return e->ctr_slow_throttles.load(std::memory_order_relaxed);
    case KT_COUNTER_PACKED_WORDS: return e->ctr_packed_words.load(std::memory_order_relaxed);
    default: return -1;
  }
}

int32_t kt_partial_layout(int32_t n_dims, int32_t* stride, int32_t* off_values, int32_t* off_presence, int32_t* off_pods,
                          int32_t* off_errors) {
  if (n_dims < 1 || n_dims > KT_MAX_DIMS) return KT_ERR_INVALID_ARGUMENT;
  if (stride) *stride = kt::partial_stride(n_dims);
  if (off_values) *off_values = 0;
  if (off_presence) *off_presence = kt::partial_off_presence(n_dims);
  if (off_pods) *off_pods = kt::partial_off_pods(n_dims);
  if (off_errors) *off_errors = kt::partial_off_errors(n_dims);
  return KT_OK;
}

const char* kt_kernel_name(kt_engine* e, int32_t kernel) {
  if (!e || kernel < 0 || kernel >= KT_KERNEL_COUNT) return "";
  return e->last_kernel[kernel];  // symbol of the kernel last dispatched for this family
}

