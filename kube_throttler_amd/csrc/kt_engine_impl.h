// kt_engine_impl.h — what the translation units of the engine's host side share: the engine object, its locks, the helpers that
// cross file boundaries.  Internal to libkt_engine.so (nothing here is part of the C-ABI; the shared helpers are hidden symbols).
//   kt_engine.cpp            create / destroy, uploads, status sync, timing, counters
//   kt_engine_compile.cpp    throttles + namespaces -> selector program + index (compile_program)
//   kt_engine_feed.cpp       state feed: namespaces, pods, throttles, status, reserved amounts, snapshots
//   kt_engine_reconcile.cpp  aggregate / exchange (kt_comm_*) / finalize and their fetches
//   kt_engine_check.cpp      PreFilter: sweeps, few-pod checks, admission queues, pages
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <chrono>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/kt_engine.h"
#include "kt_index.h"
#include "kt_launch.h"


#define KT_INTERNAL __attribute__((visibility("hidden")))

namespace kte {


struct Req {
  uint8_t op;
  uint32_t key;
  std::vector<uint32_t> vals;
};
struct Term {
  uint8_t flags = 0;
  std::vector<Req> preq, nreq;
};
struct HostAmount {
  int64_t v[KT_MAX_DIMS] = {0};
  int64_t v_hi[KT_MAX_DIMS] = {0};  // status.used only: high 64 bits of a sum beyond int64 (else the sign extension of v)
  uint32_t present = 0;
  int64_t count = 0;
  uint8_t has_count = 0;
};
struct Override {
  int64_t begin_s, end_s;
  int32_t begin_ns, end_ns;
  uint8_t flags;
  HostAmount thr;
};
struct HostThrottle {
  uint32_t flags = 0;  // KT_THR_* (0 = empty row)
  uint32_t ns = 0;
  HostAmount spec, calc, used, reserved;
  uint32_t thrl_flag = 0, thrl_has = 0;
  uint64_t status_fp = 0, spec_fp = 0;
  std::vector<Override> ovr;
  std::vector<Term> terms;
  // namespace side of the terms, evaluated once per (throttle, namespace generation): bit n of row k = term k can apply
  // to pods of namespace n.  A throttle event then costs the evaluation of ONE throttle's namespaceSelectors, not of all
  std::vector<uint32_t> adm;
  uint64_t adm_gen = 0;
  uint32_t adm_ns = 0;
};
// what the selector program and the index are compiled from
static bool same_reqs(const std::vector<Req>& a, const std::vector<Req>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i].op != b[i].op || a[i].key != b[i].key || a[i].vals != b[i].vals) return false;
  return true;
}
static bool same_selector(const HostThrottle& a, const HostThrottle& b) {
  const uint32_t sel = KT_THR_VALID | KT_THR_RESPONSIBLE | KT_THR_CLUSTER;
  if ((a.flags & sel) != (b.flags & sel) || a.ns != b.ns || a.terms.size() != b.terms.size()) return false;
  for (size_t k = 0; k < a.terms.size(); ++k)
    if (a.terms[k].flags != b.terms[k].flags || !same_reqs(a.terms[k].preq, b.terms[k].preq) || !same_reqs(a.terms[k].nreq, b.terms[k].nreq))
      return false;
  return true;
}
struct HostNamespace {
  bool valid = false;
  std::vector<std::pair<uint32_t, uint32_t>> labels;  // (key id, pair id)
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(n, 16);
    hipError_t e = kt::kt_alloc_device((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct AmountDev {
  DevBuf<int64_t> v;
  DevBuf<uint32_t> present;
  DevBuf<int64_t> count;
  DevBuf<uint8_t> has_count;
  hipError_t reserve(size_t n, int D) {
    hipError_t e;
    if ((e = v.reserve(n * D)) != hipSuccess) return e;
    if ((e = present.reserve(n)) != hipSuccess) return e;
    if ((e = count.reserve(n)) != hipSuccess) return e;
    return has_count.reserve(n);
  }
  kt::AmountTab tab() const { return kt::AmountTab{v.p, present.p, count.p, has_count.p}; }
  void release() { v.release(); present.release(); count.release(); has_count.release(); }
};

struct AmountHostFlat {
  std::vector<int64_t> v;
  std::vector<uint32_t> present;
  std::vector<int64_t> count;
  std::vector<uint8_t> has_count;
  void resize(size_t n, int D) {
    v.assign(n * D, 0);
    present.assign(n, 0);
    count.assign(n, 0);
    has_count.assign(n, 0);
  }
  void set(size_t i, int D, const HostAmount& a) {
    for (int d = 0; d < D; ++d) v[i * D + d] = (a.present >> d) & 1u ? a.v[d] : 0;
    present[i] = a.present;
    count[i] = a.has_count ? a.count : 0;
    has_count[i] = a.has_count;
  }
  void get(size_t i, int D, HostAmount& a) const {
    for (int d = 0; d < D; ++d) a.v[d] = v[i * D + d];
    a.present = present[i];
    a.count = count[i];
    a.has_count = has_count[i];
  }
};

struct TimingFamily {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
};

}  // namespace kte
using namespace kte;

extern thread_local std::string g_create_error;  // text of the last kt_engine_create failure on this thread


// The A/B switches of the engine (environment variables, all off by default): read ONCE per engine — at kt_engine_create
// and again on kt_debug_reload_env, which tools/latency_bench.py calls after it flips one on a live engine — instead of by
// getenv on every pod event and launch (getenv is not safe beside a setenv of another thread, and the pod event
// path is tuned to a few microseconds).
enum EnvSwitch { kSw_FEED_NO_STAGE, kSw_FORCE_NS_ORDER, kSw_INGEST_EVENT_WAIT, kSw_NO_FEED_FEW, kSw_NO_FEED_FUSION, kSw_NO_FUSED, kSw_NO_NS_ORDER, kSw_NO_PACK, kSw_NO_SCAN_VIEW, kSw_NO_SWEEP, kSw_NO_VERDICT_IMAGES, kSw_NO_WG_RANGES, kSw_SYNC_INGEST, kSw_INGEST_TRUST_FENCE, kSw_NO_VIEW_PATCH, kSw_CHECK_ONE_PER_CU, kSw_AGG_SMALL_WINDOW, kSwCount };
static const char* const kEnvSwitchName[kSwCount] = {"KT_FEED_NO_STAGE", "KT_FORCE_NS_ORDER", "KT_INGEST_EVENT_WAIT", "KT_NO_FEED_FEW", "KT_NO_FEED_FUSION", "KT_NO_FUSED", "KT_NO_NS_ORDER", "KT_NO_PACK", "KT_NO_SCAN_VIEW", "KT_NO_SWEEP", "KT_NO_VERDICT_IMAGES", "KT_NO_WG_RANGES", "KT_SYNC_INGEST", "KT_INGEST_TRUST_FENCE", "KT_NO_VIEW_PATCH", "KT_CHECK_ONE_PER_CU", "KT_AGG_SMALL_WINDOW"};
struct kt_engine {
  bool sw[kSwCount] = {};  // EnvSwitch values (load_env_switches)
  kt_config cfg{};
  // writers (state feed, launches, fetches) hold it exclusively for the duration of the call; the single-pod PreFilter
  // path (kt_check with n <= 8, summaries only) holds it SHARED: it reads device tables nobody may rewrite meanwhile,
  // but it does not queue behind the kernels a reconcile launch left running (controller.go:52-62: PreFilter reads
  // RW-safe caches while the reconcile workers run)
  std::shared_mutex mu;
  std::mutex small_mu;  // serialises the few-pod callers among themselves (one scratch / pinned slot)
  std::mutex ingest_mu; // settle_ingest
  // Every call except the few-pod check takes op_mu first: among themselves those calls are serialised exactly as under
  // the single mutex of rounds 1-2 (every interleaving equals some serial order).  What they take of `mu` depends on what
  // they do to the state a few-pod check reads (pod tables, selector program + index, namespace table, CheckRecs):
  //   state feed (upserts, deletes, snapshots, status / reserved amounts)      -> exclusive
  //   launches, fetches, timing — they only enqueue kernels and move results   -> shared (exclusive when the first call
  //                                                                              after a state change has to recompile /
  //                                                                              re-upload: ensure_ready)
  // so a PreFilter call waits for a pod informer event, not for a reconcile worker's launch or fetch.
  std::mutex op_mu;
  std::mutex recs_mu;  // the CheckRec bookkeeping below (launches update it under the shared lock, the few-pod path reads it)
  std::mutex err_mu;
  std::string err;
  int device = 0;
  hipStream_t own_stream = nullptr;
  int D = 0, L = 0;

  // ---- pods (device only)
  kt::PodTable pods{};
  int64_t pod_rows_hi = 0;             // 1 + highest row ever upserted
  unsigned __int128 max_abs[KT_MAX_DIMS] = {0};  // max |effective request| bound per dimension
  uint64_t or_abs[KT_MAX_DIMS] = {0};            // OR of every |request| fed: its trailing zero bits are common to all of them
  kt::PackPlan pack;                             // packed fold of the current scan view (nw == 0: plain fold)
  std::vector<unsigned long long> h_ns_end;      // host copy of the namespace ends of a namespace-ordered list (plan_wg_ranges)
  std::vector<uint32_t> h_range;                 // ... and the ranges planned from it, on their way to the device
  bool cut_plain = false;                        // a scan needed the plain fold: the index chunks stay cut for plain records
  void* cur_launch_lock = nullptr;               // the LaunchLock of the launch-side call in progress (set and cleared under op_mu)
  std::atomic<int64_t> ctr_index_chunks{0}, ctr_index_words{0}, ctr_index_image_words{0}, ctr_ns_rows{0}, ctr_ns_word_visits{0}, ctr_ns_chunk_visits{0}, ctr_slow_throttles{0}, ctr_packed_words{0};
  DevBuf<uint64_t> d_vc_pk;                      // packed request words of the countable list, scan order
  DevBuf<uint16_t> d_latom;                      // pods.latom: rewritten per selector program (kt_translate_pods)
  DevBuf<unsigned long long> d_overflow;         // valid pods whose relevant atoms did not fit pods.LA
  unsigned long long n_overflow = 0;
  // Pod events without a stream synchronisation (round 4): a small batch is packed into one of kEvSlots pinned slots the
  // kernels read directly, an event is recorded behind its kernels and the call returns; upserts / deletes pipeline on the
  // engine's stream, EVERY other entry point first waits for the last event (settle_ingest) — it is then as if the feed
  // calls had blocked themselves, which is what they did up to round 3.
  static constexpr int kEvSlots = 8;
  static constexpr size_t kEvSlotBytes = 64 * 1024;
  struct EvSlot {
    uint8_t* h = nullptr;
    hipEvent_t ev = nullptr;
    bool used = false;
  } ev_slots[kEvSlots];
  int ev_next = 0;
  std::atomic<bool> ingest_pending{false};
  hipEvent_t ingest_ev = nullptr;            // the event behind the newest asynchronous feed call
  std::atomic<hipEvent_t> ingest_unretired{nullptr};  // ... while its kernel may not have retired yet (settle_ingest returned on the spin)
  hipStream_t ingest_stream = nullptr;       // the stream the feed kernels run on
  unsigned long long* h_overflow = nullptr;  // pinned: n_overflow as the newest asynchronous translate left it; word 1: the
                                             // sequence number of the last event kernel (kt_feed_small / kt_unfeed_small) that finished
  bool overflow_in_flight = false;
  unsigned long long ingest_seq = 0;         // sequence numbers handed to the event kernels
  unsigned long long ingest_spin_seq = 0;    // != 0: the newest asynchronous feed call signals h_overflow[1] = this (settle_ingest spins)
  DevBuf<int64_t> d_countable;                   // rows of the pods a reconcile scans (kt_compact_countable)
  DevBuf<unsigned long long> d_n_countable;
  unsigned long long n_countable = 0;
  bool req_sums_valid = true;                    // the requests of the current pods are proven to add up inside 2^60
  unsigned __int128 req_sum_bound[KT_MAX_DIMS] = {0};  // >= sum of |request| over the pods held, per dimension: the last exact
                                                 // device total + everything fed since (overwritten / deleted pods stay in)
  DevBuf<unsigned long long> d_req_sums;
  // Pod events are applied to the scan lists / views IN PLACE (kt_patch_scan_views) as long as they fit what the views were
  // built for; d_pos_c / d_pos_a map a pod row to its record.  view_cap_c: records the countable view holds; view_extra:
  // upper bound of the records appended since it was built (the scan covers n_countable + view_extra records: what was
  // not really appended is zero = not countable); view_check_dirty: a namespace-ordered view was patched — the kernel
  // raises d_view_dirty when an entry would have had to move, read before the next scan
  DevBuf<int32_t> d_pos_c, d_pos_a;
  DevBuf<uint32_t> d_view_dirty;
  DevBuf<unsigned long long> d_n_all;
  int64_t view_cap_c = 0, view_extra = 0, view_rows_a = 0;
  bool view_check_dirty = false;
  bool countable_valid = false;                  // d_countable describes the current pod table
  bool countable_by_ns = false;                  // ... ordered by namespace (multi-chunk index: kt_order_rows_by_ns)
  DevBuf<int64_t> d_order_all;                   // every pod row ordered by namespace: the check sweep's scan order
  bool order_all_valid = false;
  DevBuf<unsigned long long> d_ns_cursor;        // counting-sort scratch (one word per namespace row)
  // record ranges of the workgroups of a namespace-ordered scan, ends at namespace boundaries (plan_wg_ranges, host side): the all-rows
  // list (check sweep) and the countable list (aggregate); *_G = the grid they were planned for (0: none)
  DevBuf<uint32_t> d_range_a, d_range_c;
  int range_a_G = 0, range_c_G = 0;
  // scan-ordered copies of the listed pods' records (kt_build_scan_view): countable list / all-rows list
  DevBuf<uint64_t> d_vc_meta, d_va_meta, d_carry;
  DevBuf<uint16_t> d_vc_latom, d_va_latom;
  DevBuf<int64_t> d_vc_req;
  DevBuf<uint8_t> d_row_mask;                    // kt_reconcile_rows_launch: the keys of the reconcile, a byte per throttle row
  DevBuf<uint32_t> d_slab_tag;                   // [chunks][256] epoch of the launch that last spilled a slab
  uint32_t slab_epoch = 0;
  bool neg_seen = false;                         // some pod was fed with a negative request (sums may cancel)

  // ---- host mirrors of the small tables
  std::vector<HostNamespace> ns;
  uint64_t ns_gen = 1;  // bumped by every namespace event (HostThrottle::adm is keyed by it)
  int32_t ns_rows_hi = 0;
  int64_t pod_ns_hi = 0;   // 1 + highest namespace row any pod was fed with
  size_t ns_compiled = 0;  // namespace rows the compiled program / index cover (compile_program)
  std::vector<HostThrottle> thr;
  int32_t thr_rows_hi = 0;
  bool program_dirty = true;   // selectors / namespaces / the set of throttle rows changed -> recompile + index + upload
  bool spec_dirty = false;     // only spec.threshold / overrides / message fingerprints of existing rows changed (the usual
                               // Throttle event: a threshold edit, the controller's own status update) -> their tables only
  bool status_host_dirty = true;  // host status/reserved rows newer than device
  bool reserved_dev_newer = false;  // device reserved rows newer than the host mirrors (admit with commit)
  bool incremental = false;         // KT_VARIANT_INCREMENTAL: `used` partials maintained by pod deltas
  bool agg_valid = false;           // d_agg = this GPU's partials for the current pods + selector program
  DevBuf<unsigned long long> d_agg;
  bool recs_valid = false;          // d_recs matches the device status + reserved tables for (recs_eq, recs_DT)
  bool recs_eq = false;
  int recs_DT = 0;
  bool status_dev_newer = false;  // device status newer than host (after reconcile with APPLY)

  // ---- compiled program (device)
  DevBuf<uint32_t> d_thr_term_off, d_term_thr, d_term_req_off, d_req_key, d_req_val_off, d_req_val, d_ns_term_ok;
  DevBuf<uint8_t> d_term_flags, d_req_op, d_ns_valid;
  kt::SelProgram sp{};
  DevBuf<kt::SelProgram> d_sp;  // device copy (kernels that touch the program only on rare paths take a pointer)
  bool uses_keys = false;
  kt::HostIndex hindex;
  kt::IndexDev dindex;

  // ---- throttle tables (device)
  DevBuf<uint32_t> d_thr_flags, d_thrl_flag, d_thrl_has, d_ovr_off;
  DevBuf<uint64_t> d_status_fp, d_spec_fp;
  AmountDev d_spec, d_calc, d_used, d_reserved, d_ovr_thr;
  DevBuf<int64_t> d_ovr_begin_s, d_ovr_end_s;
  DevBuf<int32_t> d_ovr_begin_ns, d_ovr_end_ns;
  DevBuf<uint8_t> d_ovr_flags;
  kt::ThrTables tt{};

  // ---- reconcile state
  DevBuf<unsigned long long> d_partial;
  DevBuf<uint8_t> d_admit;  // HBM-resident state of kt_admit_sequential when it does not fit LDS
  DevBuf<uint8_t> d_slab;  // per-workgroup LDS table spill area of kt_aggregate_bitmap
  unsigned long long* ext_partial = nullptr;  // caller-owned partial buffer (kt_use_partial_buffer)
  int64_t ext_partial_words = 0;
  unsigned long long* partial() { return ext_partial ? ext_partial : d_partial.p; }
  const void* clean_partial = nullptr;  // the partial buffer known to hold zeros (left behind by a consuming finalize)
  AmountDev d_out_used, d_out_calc;
  // wide sums: when the requests of the pods held add up beyond int64 a reconcile scans twice (low 32-bit limbs, the rest)
  // and kt_finalize joins the sums in 128 bits; the high words of `used` live beside the int64 tables
  bool wide = false;          // decided by request_sums_in_range
  bool agg_wide = false;      // the pending partials are limb sums: [2][T][2D+2]
  DevBuf<int64_t> d_used_hi, d_out_used_hi;
  DevBuf<uint8_t> d_out_calc_updated, d_out_thrl_pod, d_out_error;
  DevBuf<int64_t> d_out_next_s;
  DevBuf<int32_t> d_out_next_ns;
  DevBuf<uint32_t> d_out_thrl_flag, d_out_thrl_has;
  bool reconcile_ready = false;
  // the partial buffer as the last kt_aggregate_launch filled it: word count and the selector program it was scanned
  // with — the exchange and the finalize that follow must see the same throttle set
  bool agg_pending = false;
  // a packed scan whose slabs still wait for kt_reduce_finalize_packed (kt_reconcile_launch: nothing can come between the
  // scan and the finalize, so the slab reduction and kt_finalize are ONE launch): workgroups of the scan, its slab epoch
  bool fused_pending = false;
  int fused_nb = 0;
  uint32_t fused_epoch = 0;
  kt::PackPlan fused_pack;  // the plan the pending slabs were written with (the aggregate's view plan, or kt_sweep_launch's own)
  size_t agg_words = 0;
  uint64_t program_gen = 0, agg_gen = 0;
  int32_t exchange_world = 1;  // ranks whose partials meet in the reconcile's all-reduce (kt_comm_init / kt_set_exchange_world)

  // ---- check state
  // CheckRecs, double-buffered: a reconcile with APPLY writes the NEW generation into the other buffer and records an
  // event behind it; until that event has completed, a concurrent single-pod check reads the previous generation (a
  // consistent status: the one before the reconcile) instead of waiting for — or racing with — kt_finalize
  DevBuf<uint8_t> d_recs2[2];
  int recs_cur = 0;
  // per CheckRecs buffer: how often it was rewritten, and the per-word check tables (TermInfo + WordVerdict of every word of
  // the index: kt_build_verdict_images) built from it — valid while (recs_seq, program_gen, DT) are those of the build
  uint64_t recs_seq[2] = {0, 0};
  DevBuf<uint8_t> d_wvimg[2];
  uint64_t wvimg_seq[2] = {~0ull, ~0ull}, wvimg_gen[2] = {~0ull, ~0ull};
  int wvimg_DT[2] = {0, 0};
  hipEvent_t recs_ev[2] = {nullptr, nullptr};
  bool recs_ev_pending[2] = {false, false};
  int32_t wide_mode = 0;  // kt_set_wide_sums: 0 = decided per engine (single rank only), 1 = always two blocks
  bool recs_prev_valid = false;  // the other buffer holds complete records of the same (program, on_equal, DT)
  uint8_t* recs_ptr() { return d_recs2[recs_cur].p; }
  // ---- few-pod check path (kt_kernels_few.hip)
  hipStream_t small_stream = nullptr;  // high priority: its one-wave workgroups start beside a running sweep
  DevBuf<unsigned long long> d_few_acc;
  DevBuf<uint32_t> d_few_ticket;
  uint64_t* h_few = nullptr;  // pinned: [8] summary words, [8] = sequence number
  uint64_t few_seq = 0;
  std::atomic<bool> few_ready{false};
  std::atomic<int64_t> few_served{0};
  std::atomic<int64_t> n_compiles{0};
  DevBuf<uint64_t> d_summary;
  DevBuf<uint8_t> d_status;
  DevBuf<int64_t> d_rows;
  int64_t check_n = 0;
  DevBuf<uint32_t> d_ticket;          // arrival counters of small check launches (kt_check_bitmap SMALL)
  uint64_t* h_small = nullptr;        // pinned host copy of a small launch's summary words (kCheckSmallMax)
  bool check_in_h_small = false;      // the last check left its summaries in h_small
  int32_t check_T = 0, reconcile_T = 0;  // throttle rows in effect when the last check / reconcile was launched
  bool check_has_status = false;
  bool check_ready = false;
  hipStream_t last_stream = nullptr;

  // ---- staging
  DevBuf<uint8_t> d_stage;
  DevBuf<uint8_t> d_ev_stage;  // kt_feed_small's device copy of a pinned event slot (kEvSlotBytes, allocated once)
  uint8_t* h_stage = nullptr;  // pinned: small batches cross in one copy

  // ---- RCCL communicator (kt_comm_*): opaque ncclComm_t, rank / world
  void* comm = nullptr;
  int32_t comm_rank = 0, comm_world = 1;

  const char* last_kernel[KT_KERNEL_COUNT] = {"", "", "kt_finalize", "kt_prepare_check", "kt_reduce_bitmap_slabs"};

  // ---- timing
  bool timing = false;
  TimingFamily fam[KT_KERNEL_COUNT];

  int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    {
      std::lock_guard<std::mutex> g(err_mu);
      err = buf;
    }
    return code;
  }
};

#define KT_HIP(e, call)                                                                          \
  do {                                                                                           \
    hipError_t _r = (call);                                                                      \
    if (_r != hipSuccess) return (e)->fail(KT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(_r)); \
  } while (0)

// ---- helpers every file uses (inline)
// what asynchronous pod feed calls left in flight: wait for it (any thread; idempotent)
inline void settle_ingest(kt_engine* e) {
  if (!e->ingest_pending.load(std::memory_order_acquire)) return;
  std::lock_guard<std::mutex> g(e->ingest_mu);
  if (!e->ingest_pending.load(std::memory_order_acquire)) return;
  (void)hipSetDevice(e->device);
  bool done = false;
  if (e->ingest_spin_seq && e->h_overflow) {
    // the event kernel stores its sequence number into pinned memory behind a system-scope release of everything it
    // wrote: a few microseconds of polling instead of hipEventSynchronize's 35-40 (the event is the fallback)
    volatile unsigned long long* sq = e->h_overflow + 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 0; !(done = *sq >= e->ingest_spin_seq); ++it) {
      __builtin_ia32_pause();
      if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done) (void)hipEventSynchronize(e->ingest_ev);
  // The spin returns when the feed kernel has STORED its sequence number (behind a system-scope release of everything it
  // wrote), not when it has retired.  Measured sufficient on gfx950 (tools/microbench/meet_litmus.hip), but HIP does not
  // promise it for coarse-grained allocations: every stream other than the feed's own is therefore also ordered behind the
  // kernel's event on the DEVICE side before it reads the pod tables (order_behind_ingest: one hipStreamWaitEvent per such
  // launch until the event has completed — no host wait).
  // (KT_INGEST_TRUST_FENCE=1 skips that ordering and relies on the measured behaviour: 35 instead of 41 us from a pod event to
  //  the PreFilter that sees it)
  e->ingest_unretired = done && !e->sw[kSw_INGEST_TRUST_FENCE] ? e->ingest_ev : nullptr;
  if (e->overflow_in_flight) e->n_overflow = *e->h_overflow, e->overflow_in_flight = false;
  e->ingest_pending.store(false, std::memory_order_release);
}
// a launch on `s` that reads what the newest feed kernel wrote: behind that kernel on the device (see settle_ingest)
inline void order_behind_ingest(kt_engine* e, hipStream_t s) {
  hipEvent_t ev = e->ingest_unretired.load(std::memory_order_acquire);
  if (!ev) return;
  if (hipEventQuery(ev) == hipSuccess) {  // retired meanwhile: nothing to order any more
    e->ingest_unretired.compare_exchange_strong(ev, nullptr);
    return;
  }
  if (s != e->ingest_stream) (void)hipStreamWaitEvent(s, ev, 0);
}
// state feed: nobody else inside
struct StateLock {
  std::unique_lock<std::mutex> op;
  std::unique_lock<std::shared_mutex> ex;
  explicit StateLock(kt_engine* e, bool settle = true) : op(e->op_mu), ex(e->mu) {
    if (settle) settle_ingest(e);
  }
};
// launches / fetches: serialised among themselves (op_mu), beside few-pod checks (shared) — unless this call will have to
// recompile or re-upload state those checks read (the dirty flags are only written under op_mu + exclusive mu, so reading
// them with op_mu held is safe)
struct LaunchLock {
  kt_engine* e;
  std::unique_lock<std::mutex> op;
  std::unique_lock<std::shared_mutex> ex;
  std::shared_lock<std::shared_mutex> sh;
  explicit LaunchLock(kt_engine* e_, bool force_exclusive = false) : e(e_), op(e_->op_mu) {
    if (force_exclusive || e->program_dirty || e->status_host_dirty) ex = std::unique_lock<std::shared_mutex>(e->mu);
    else sh = std::shared_lock<std::shared_mutex>(e->mu);
    e->cur_launch_lock = this;  // (op_mu is held: one launch-side call at a time)
    settle_ingest(e);
  }
  ~LaunchLock() { e->cur_launch_lock = nullptr; }
  // a launch that finds it has to change state a few-pod check reads after all (the index cut again for plain records):
  // shared -> exclusive.  op_mu stays held, so no other launch / feed call comes between; few-pod checks may.
  void upgrade() {
    if (!sh.owns_lock()) return;
    sh.unlock();
    ex = std::unique_lock<std::shared_mutex>(e->mu);
  }
};
struct TimedLaunch {
  kt_engine* e;
  int family;
  hipStream_t s;
  hipEvent_t stop = nullptr;
  TimedLaunch(kt_engine* e_, int family_, hipStream_t s_) : e(e_), family(family_), s(s_) {
    if (!e->timing) return;
    TimingFamily& f = e->fam[family];
    if (f.used == f.pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      f.pool.emplace_back(a, b);
    }
    auto& pr = f.pool[f.used++];
    (void)hipEventRecord(pr.first, s);
    stop = pr.second;
  }
  void stop_now() {
    if (stop) (void)hipEventRecord(stop, s);
    stop = nullptr;
  }
  ~TimedLaunch() { stop_now(); }
};


template <class T>
int32_t upload(kt_engine* e, DevBuf<T>& d, const std::vector<T>& h, hipStream_t s) {
  KT_HIP(e, d.reserve(h.size() + 1));
  if (!h.empty()) KT_HIP(e, hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return KT_OK;
}

constexpr unsigned __int128 kSumBound = (unsigned __int128)1 << 60;
// Per-rank bound of the summed |request| per dimension: the all-reduced `used` of `world` ranks must stay inside int64
// (kt_finalize reads it as int64), so 2^60 up to 4 ranks and 2^62 / world (rounded down to a power of two) beyond.
inline unsigned __int128 rank_sum_bound(int32_t world) {
  unsigned __int128 b = kSumBound;
  for (int32_t w = 4; w < world; w *= 2) b >>= 1;
  return b;
}
// Between kt_aggregate_launch and the calls that consume its partials (kt_comm_allreduce_partial, kt_finalize_launch) the
// throttle set must not change: a grown thr_rows_hi would read past the buffer the scan filled, ranks would disagree on
// the word count, and ensure_ready would recompile and clear the buffer.
#define KT_CHECK_PARTIALS_CURRENT(e, who)                                                                              \
  do {                                                                                                                 \
    if ((e)->agg_pending && ((e)->program_dirty || (e)->agg_gen != (e)->program_gen ||                                 \
                             (e)->agg_words != (size_t)(e)->thr_rows_hi * kt::partial_stride((e)->D) * ((e)->agg_wide ? 2u : 1u)))                 \
      return (e)->fail(KT_ERR_NOT_READY, who ": throttles or namespaces changed since kt_aggregate_launch filled the "  \
                                             "partial buffer; aggregate again");                                       \
  } while (0)
constexpr size_t kPinnedStageBytes = 1u << 20;

inline unsigned __int128 uabs(int64_t x) { return x < 0 ? (unsigned __int128)(-(__int128)x) : (unsigned __int128)x; }

// RCCL, loaded on first use (kt_engine_reconcile.cpp)
struct Rccl {
  struct Id128 {  // ncclUniqueId: passed BY VALUE to ncclCommInitRank
    char b[128];
  };
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
KT_INTERNAL Rccl* rccl();

// ---- helpers that cross file boundaries (hidden symbols)
KT_INTERNAL void recs_invalidate_and_drain(kt_engine* e);
KT_INTERNAL hipStream_t pick_stream(kt_engine* e, void* s);
KT_INTERNAL int32_t upload_amounts(kt_engine* e, AmountDev& d, const AmountHostFlat& h, size_t n, int D, hipStream_t s);
KT_INTERNAL int32_t download_amounts(kt_engine* e, const AmountDev& d, AmountHostFlat& h, size_t n, int D, hipStream_t s);
KT_INTERNAL int32_t sync_status_to_host(kt_engine* e);
KT_INTERNAL int32_t upload_status(kt_engine* e, hipStream_t s);
KT_INTERNAL int32_t upload_spec_tables(kt_engine* e, hipStream_t s);
KT_INTERNAL int32_t compile_program(kt_engine* e, hipStream_t s);
KT_INTERNAL int32_t ensure_ready(kt_engine* e, hipStream_t s);
KT_INTERNAL void amount_from_table(const kt_amounts& a, size_t i, int D, HostAmount& h);
KT_INTERNAL void amount_to_table(const HostAmount& h, const kt_amounts& a, size_t i, int D);
KT_INTERNAL kt::ReqBound req_bound(const kt_engine* e);
KT_INTERNAL bool amount_in_bound(const HostAmount& a, int D);
KT_INTERNAL void reqs_from_pool(const kt_reqs& pool, uint32_t b, uint32_t e_, std::vector<Req>& out);
// (kt_engine_feed.cpp) before a scan uses a namespace-ordered view that was patched: did an entry have to move?
KT_INTERNAL int32_t settle_view_patches(kt_engine* e, hipStream_t s);
// (kt_engine_reconcile.cpp)
KT_INTERNAL int32_t slab_tags(kt_engine* e, kt::AggScan& sc, hipStream_t s);
KT_INTERNAL int32_t request_sums_in_range(kt_engine* e, hipStream_t s);
KT_INTERNAL int32_t aggregate_locked(kt_engine* e, hipStream_t s, bool allow_fused = false);
KT_INTERNAL int32_t delta_scan(kt_engine* e, int64_t n, const int64_t* rows_dev, int64_t row0, int sign, hipStream_t s);
KT_INTERNAL int32_t finalize_locked(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, hipStream_t s, bool consume = false,
                                    const uint8_t* row_mask = nullptr);

