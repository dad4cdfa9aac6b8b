// kt_engine.cpp — host side of libkt_engine.so: the C-ABI of include/kt_engine.h over the HIP kernels.
//
// Responsibilities: own every device allocation (pod row tables, throttle tables, selector program,
// index, result buffers), validate and stage caller batches (the caller's memory is never retained),
// compile throttles + namespaces into the device selector program, launch kernels on the caller's
// stream, time them with HIP events.  No compute happens here: without a gfx950 device
// kt_engine_create fails (KT_ERR_NO_DEVICE) — there is no CPU fallback.
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

namespace {

thread_local std::string g_create_error;

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

}  // namespace

// The A/B switches of the engine (environment variables, all off by default): read ONCE per engine — at kt_engine_create
// and again on kt_debug_reload_env, which tools/latency_bench.py calls after it flips one on a live engine — instead of by
// getenv on every pod event and launch (getenv is not safe beside a setenv of another thread, and the pod event
// path is tuned to a few microseconds).
enum EnvSwitch { kSw_FEED_NO_STAGE, kSw_FORCE_NS_ORDER, kSw_INGEST_EVENT_WAIT, kSw_NO_FEED_FEW, kSw_NO_FEED_FUSION, kSw_NO_FUSED, kSw_NO_NS_ORDER, kSw_NO_PACK, kSw_NO_SCAN_VIEW, kSw_NO_SWEEP, kSw_NO_VERDICT_IMAGES, kSw_NO_WG_RANGES, kSw_SYNC_INGEST, kSw_INGEST_TRUST_FENCE, kSwCount };
static const char* const kEnvSwitchName[kSwCount] = {"KT_FEED_NO_STAGE", "KT_FORCE_NS_ORDER", "KT_INGEST_EVENT_WAIT", "KT_NO_FEED_FEW", "KT_NO_FEED_FUSION", "KT_NO_FUSED", "KT_NO_NS_ORDER", "KT_NO_PACK", "KT_NO_SCAN_VIEW", "KT_NO_SWEEP", "KT_NO_VERDICT_IMAGES", "KT_NO_WG_RANGES", "KT_SYNC_INGEST", "KT_INGEST_TRUST_FENCE"};
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
  std::atomic<int64_t> ctr_index_chunks{0}, ctr_index_words{0}, ctr_index_image_words{0}, ctr_ns_rows{0}, ctr_ns_word_visits{0}, ctr_ns_chunk_visits{0}, ctr_slow_throttles{0};
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
  // record ranges of the workgroups of a namespace-ordered scan, ends at namespace boundaries (kt_plan_wg_ranges): the all-rows
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

namespace {

static bool getenv_flag(const char* name) {
  const char* v = getenv(name);
  return v && *v && *v != '0';
}
static void load_env_switches(kt_engine* e) {
  for (int k = 0; k < kSwCount; ++k) e->sw[k] = getenv_flag(kEnvSwitchName[k]);
}
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

// ---- host-side label selector evaluation (namespace selectors only; pods are matched on device) -----
bool ns_selector_matches(const std::vector<Req>& reqs, const HostNamespace& n) {
  for (const Req& r : reqs) {
    bool has = false, in = false;
    for (auto& kv : n.labels) {
      if (kv.first == r.key) has = true;
      for (uint32_t v : r.vals) in |= kv.second == v;
    }
    bool ok;
    switch (r.op) {
      case KT_OP_IN: ok = in; break;
      case KT_OP_NOT_IN: ok = !in; break;
      case KT_OP_EXISTS: ok = has; break;
      case KT_OP_DOES_NOT_EXIST: ok = !has; break;
      default: ok = false;
    }
    if (!ok) return false;
  }
  return true;
}

template <class T>
int32_t upload(kt_engine* e, DevBuf<T>& d, const std::vector<T>& h, hipStream_t s) {
  KT_HIP(e, d.reserve(h.size() + 1));
  if (!h.empty()) KT_HIP(e, hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return KT_OK;
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

int32_t compile_program(kt_engine* e, hipStream_t s) {
  static const bool dbg_time = getenv("KT_DEBUG_COMPILE") != nullptr;  // phase times of a recompile on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "compile_program: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const int D = e->D;
  const size_t T = (size_t)e->thr_rows_hi;
  // namespace rows the program covers: the rows in USE (namespace objects, pods, namespaced Throttles), not the
  // configured capacity — the per-namespace tables of the program and of every index chunk scale with this number.
  // A pod that later arrives with a higher namespace row marks the program dirty (upsert_pods_locked).
  size_t NS = std::max<size_t>(1, std::max<size_t>((size_t)e->ns_rows_hi, (size_t)e->pod_ns_hi));
  for (size_t t = 0; t < T; ++t)
    if ((e->thr[t].flags & KT_THR_VALID) && !(e->thr[t].flags & KT_THR_CLUSTER)) NS = std::max(NS, (size_t)e->thr[t].ns + 1);
  NS = std::min(NS, (size_t)e->cfg.namespace_capacity);
  e->ns_compiled = NS;
  std::vector<uint32_t> thr_term_off(T + 1, 0), term_thr, term_req_off{0}, req_key, req_val_off{0}, req_val;
  std::vector<uint8_t> term_flags, req_op;
  {
    size_t n_terms = 0, n_reqs = 0, n_vals = 0;
    for (size_t t = 0; t < T; ++t)
      if (e->thr[t].flags & KT_THR_VALID)
        for (const Term& tm : e->thr[t].terms) {
          ++n_terms, n_reqs += tm.preq.size();
          for (const Req& r : tm.preq) n_vals += r.vals.size();
        }
    term_thr.reserve(n_terms), term_flags.reserve(n_terms), term_req_off.reserve(n_terms + 1);
    req_op.reserve(n_reqs), req_key.reserve(n_reqs), req_val_off.reserve(n_reqs + 1), req_val.reserve(n_vals);
  }
  e->uses_keys = false;
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    if (h.flags & KT_THR_VALID)
      for (const Term& tm : h.terms) {
        term_thr.push_back((uint32_t)t);
        term_flags.push_back(tm.flags);
        for (const Req& r : tm.preq) {
          req_op.push_back(r.op);
          req_key.push_back(r.key);
          for (uint32_t v : r.vals) req_val.push_back(v);
          req_val_off.push_back((uint32_t)req_val.size());
          if (r.op == KT_OP_EXISTS || r.op == KT_OP_DOES_NOT_EXIST) e->uses_keys = true;
        }
        term_req_off.push_back((uint32_t)req_op.size());
      }
    thr_term_off[t + 1] = (uint32_t)term_thr.size();
  }
  lap("flatten throttles");
  const size_t G = term_thr.size();
  const uint32_t gw = (uint32_t)((G + 31) / 32 + 1);
  // ns x term applicability: the namespace side of every term, cached per throttle (HostThrottle::adm) and re-evaluated
  // only for throttles that changed since — or for all of them, on several host threads, after a namespace event
  const uint32_t nsw = (uint32_t)((NS + 31) / 32);
  std::vector<uint8_t> ns_valid(NS, 0);
  for (size_t n = 0; n < NS; ++n) ns_valid[n] = n < e->ns.size() && e->ns[n].valid;
  {
    std::vector<uint32_t> stale;
    for (size_t t = 0; t < T; ++t) {
      const HostThrottle& h = e->thr[t];
      if ((h.flags & KT_THR_VALID) && (h.adm_gen != e->ns_gen || h.adm_ns != (uint32_t)NS || h.adm.size() != h.terms.size() * nsw)) stale.push_back((uint32_t)t);
    }
    kt::parallel_for(stale.size(), 64, [&](size_t b0, size_t b1, size_t) {
      for (size_t q = b0; q < b1; ++q) {
        HostThrottle& h = e->thr[stale[q]];
        h.adm.assign(h.terms.size() * nsw, 0u);
        h.adm_gen = e->ns_gen, h.adm_ns = (uint32_t)NS;
        const uint32_t need = KT_THR_VALID | KT_THR_RESPONSIBLE;
        if ((h.flags & need) != need) continue;
        for (size_t k = 0; k < h.terms.size(); ++k) {
          const Term& tm = h.terms[k];
          uint32_t* row = h.adm.data() + k * nsw;
          if (!(h.flags & KT_THR_CLUSTER)) {
            // Throttles(pod.Namespace).List: implicit namespace equality, no Namespace object needed
            if (h.ns < NS) row[h.ns >> 5] |= 1u << (h.ns & 31);
          } else {
            if (tm.flags & KT_TERM_NS_SEL_INVALID) continue;  // swallowed to "no match" (clusterthrottle_selector.go:63-69)
            for (size_t n = 0; n < NS && n < e->ns.size(); ++n)
              if (e->ns[n].valid && ns_selector_matches(tm.nreq, e->ns[n])) row[n >> 5] |= 1u << (n & 31);
          }
        }
      }
    }, nullptr);
  }
  std::vector<uint32_t> adm_all(G * nsw, 0u);  // [term][namespace words]: what the index build wants
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    if (!(h.flags & KT_THR_VALID) || h.terms.empty()) continue;
    memcpy(adm_all.data() + (size_t)thr_term_off[t] * nsw, h.adm.data(), h.adm.size() * 4);
  }
  std::vector<uint32_t> ns_term_ok;  // [namespace][term words]: what the kernels' rare paths and the dense variant read
  kt::transpose_term_ns_bits(adm_all, G, nsw, (uint32_t)NS, gw, ns_term_ok);
  lap("namespace side of the terms");
  int32_t rc;
#define UP(dev, host) if ((rc = upload(e, e->dev, host, s)) != KT_OK) return rc
  UP(d_thr_term_off, thr_term_off);
  UP(d_term_thr, term_thr);
  UP(d_term_flags, term_flags);
  UP(d_term_req_off, term_req_off);
  UP(d_req_op, req_op);
  UP(d_req_key, req_key);
  UP(d_req_val_off, req_val_off);
  UP(d_req_val, req_val);
  UP(d_ns_term_ok, ns_term_ok);
  UP(d_ns_valid, ns_valid);
#undef UP
  if ((rc = upload_spec_tables(e, s)) != KT_OK) return rc;
  // result / scratch buffers sized by T
  KT_HIP(e, e->d_partial.reserve(2 * T * kt::partial_stride(D) + 1));  // room for the two limb-sum blocks of a wide reconcile
  e->clean_partial = nullptr;
  KT_HIP(e, e->d_out_used.reserve(T + 1, D));
  KT_HIP(e, e->d_out_used_hi.reserve((T + 1) * (size_t)D));
  KT_HIP(e, e->d_out_calc.reserve(T + 1, D));
  KT_HIP(e, e->d_out_calc_updated.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_pod.reserve(T + 1));
  KT_HIP(e, e->d_out_error.reserve(T + 1));
  KT_HIP(e, e->d_out_next_s.reserve(T + 1));
  KT_HIP(e, e->d_out_next_ns.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_flag.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_has.reserve(T + 1));
  KT_HIP(e, e->d_recs2[0].reserve(kt::recs_bytes((int)T)));
  KT_HIP(e, e->d_recs2[1].reserve(kt::recs_bytes((int)T)));
  e->recs_prev_valid = false;
  lap("uploads + buffers");
  // index for the work ~ (pods + matches) kernels
  {
    auto thr_info = [&](uint32_t t) {
      const HostThrottle& h = e->thr[t];
      const uint32_t need = KT_THR_VALID | KT_THR_RESPONSIBLE;
      kt::ThrInfo ti;
      ti.live = (h.flags & need) == need;
      ti.cluster = (h.flags & KT_THR_CLUSTER) != 0;
      ti.ns = h.ns;
      return ti;
    };
    // KT_CHUNK_BUDGET (bytes): test hook that forces small chunks so that tiny programs exercise the multi-chunk path too
    const char* hook = getenv("KT_CHUNK_BUDGET");
    const uint32_t lds_all = 160u * 1024u;
    const uint32_t agg_budget = hook ? (uint32_t)atoi(hook) : lds_all - kt::aggregate_fixed_lds();
    const uint32_t thr_bytes = kt::agg_rec_bytes(D, e->incremental);
    // a program of several chunks is cut for the packed fold's records (at most 40 bytes: PackPlan) while this engine's scans
    // can pack — a chunk then holds more words, a namespace-ordered scan makes fewer chunk passes; the first scan that needs
    // the plain fold (a negative request, sums beyond int64, KT_NO_PACK) has the program cut again for plain records
    // (aggregate_locked: cut_plain)
    const uint32_t thr_packed = (!e->incremental && !e->wide && !e->neg_seen && !e->cut_plain && !e->sw[kSw_NO_PACK]) ? kt::kPackedRecMax : 0u;
    // the check kernel runs two workgroups per CU when the whole program fits half the LDS; otherwise the chunks are cut
    // for one workgroup per CU (fewer, larger chunks)
    const uint32_t chk_half = hook ? (uint32_t)atoi(hook) : lds_all / 2 - kt::check_fixed_lds();
    // KT_CHUNK_HALF=1 (A/B runs): keep the half-LDS chunks — more of them, but two workgroups per CU
    const bool full_when_chunked = !hook && !getenv("KT_CHUNK_HALF");
    kt::build_index(e->hindex, thr_term_off, term_thr, term_flags, term_req_off, req_op, req_key, req_val_off, req_val, thr_info,
                    (uint32_t)NS, ns_term_ok, gw, agg_budget, chk_half, thr_bytes, e->L, &adm_all,
                    full_when_chunked ? lds_all - kt::check_fixed_lds() : 0u, kt::check_word_lds(D), nullptr, thr_packed);
    lap("build_index");
    // a program that fits half the LDS as rows but still came out in several chunks (per-term tables): larger chunks
    if (full_when_chunked && e->hindex.bm_chunks.size() > 1 && e->hindex.cut_chk_budget != lds_all - kt::check_fixed_lds())
      kt::cut_chunks(e->hindex, agg_budget, lds_all - kt::check_fixed_lds(), thr_bytes, kt::check_word_lds(D), thr_packed);
    lap("cut_chunks (full LDS)");
  }
  kt::index_group_counts(e->hindex, (uint32_t)T);
  e->ctr_index_chunks.store((int64_t)e->hindex.bm_chunks.size(), std::memory_order_relaxed);
  e->ctr_index_words.store((int64_t)e->hindex.bm_words, std::memory_order_relaxed);
  e->ctr_index_image_words.store((int64_t)e->hindex.img_words, std::memory_order_relaxed);
  e->ctr_ns_rows.store((int64_t)e->hindex.n_ns, std::memory_order_relaxed);
  e->ctr_ns_word_visits.store(e->hindex.ns_word_visits, std::memory_order_relaxed);
  e->ctr_ns_chunk_visits.store(e->hindex.ns_chunk_visits, std::memory_order_relaxed);
  e->ctr_slow_throttles.store((int64_t)e->hindex.slow_thr.size(), std::memory_order_relaxed);
  e->agg_valid = false;  // a new selector program: the maintained partials are void
  KT_HIP(e, e->d_slab.reserve((size_t)e->hindex.bm_slab_bytes + 64));
  {
    hipError_t he = kt::upload_index(e->hindex, e->dindex, s);
    if (he != hipSuccess) return e->fail(KT_ERR_DEVICE, "upload_index: %s", hipGetErrorString(he));
  }
  lap("upload_index");
  // the pods' labels as atom ids of THIS program (labels no selector mentions drop out here)
  e->pods.LA = (int32_t)e->hindex.la;
  KT_HIP(e, e->d_latom.reserve((size_t)e->cfg.pod_capacity * (size_t)e->pods.LA + 64));
  e->pods.latom = e->d_latom.p;
  KT_HIP(e, e->d_overflow.reserve(1));
  KT_HIP(e, hipMemsetAsync(e->d_overflow.p, 0, 8, s));
  kt::launch_translate_pods(e->pods, e->pod_rows_hi, nullptr, 0, e->dindex, e->d_overflow.p, s);
  KT_HIP(e, hipGetLastError());
  e->countable_valid = false, e->order_all_valid = false;  // the scan views hold copies of the atom rows
  KT_HIP(e, hipMemcpyAsync(&e->n_overflow, e->d_overflow.p, 8, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));  // host vectors go out of scope
  e->sp.thr_term_off = e->d_thr_term_off.p;
  e->sp.term_thr = e->d_term_thr.p;
  e->sp.term_flags = e->d_term_flags.p;
  e->sp.term_req_off = e->d_term_req_off.p;
  e->sp.req_op = e->d_req_op.p;
  e->sp.req_key = e->d_req_key.p;
  e->sp.req_val_off = e->d_req_val_off.p;
  e->sp.req_val = e->d_req_val.p;
  e->sp.ns_term_ok = e->d_ns_term_ok.p;
  e->sp.ns_valid = e->d_ns_valid.p;
  e->sp.gw = gw;
  e->sp.T = (int32_t)T;
  e->sp.G = (int32_t)G;
  e->sp.n_ns = (int32_t)NS;
  KT_HIP(e, e->d_sp.reserve(1));
  KT_HIP(e, hipMemcpyAsync(e->d_sp.p, &e->sp, sizeof(kt::SelProgram), hipMemcpyHostToDevice, s));
  KT_HIP(e, hipStreamSynchronize(s));
  lap("translate pods + sync");
  e->program_dirty = false;
  ++e->program_gen;
  e->n_compiles.fetch_add(1, std::memory_order_relaxed);
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

// upper bound of every pod's effective request per dimension, for kRecTight (kt_device.h)
kt::ReqBound req_bound(const kt_engine* e) {
  kt::ReqBound b;
  for (int d = 0; d < 16; ++d)
    b.v[d] = d < e->D ? (e->max_abs[d] > (unsigned __int128)INT64_MAX ? INT64_MAX : (int64_t)e->max_abs[d]) : 0;
  return b;
}
inline unsigned __int128 uabs(int64_t x) { return x < 0 ? (unsigned __int128)(-(__int128)x) : (unsigned __int128)x; }

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

}  // namespace

namespace {
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
constexpr int kNcclInt64 = 4, kNcclSum = 0;  // rccl.h: ncclInt64, ncclSum
}  // namespace

// ===================================================================================================
// C-ABI
// ===================================================================================================
extern "C" {

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

static int32_t delta_scan(kt_engine* e, int64_t n, const int64_t* rows_dev, int64_t row0, int sign, hipStream_t s);

// ---- pod events applied to the scan views in place
constexpr int64_t kPatchBatchMax = 65536;
// can a batch of n pod rows (largest |request| per dimension batch_max, OR of the values batch_or, a negative value seen)
// be applied to the current views?  The packed request words only hold what their plan was proved for.
static bool views_patchable(const kt_engine* e, int64_t n, const unsigned __int128* batch_max, const uint64_t* batch_or, bool batch_neg) {
  if (e->incremental || e->cfg.kernel_variant != 0 || e->program_dirty || n > kPatchBatchMax) return false;
  if (!e->countable_valid && !e->order_all_valid) return false;  // nothing to patch: the next scan builds anyway
  if (getenv("KT_NO_VIEW_PATCH")) return false;
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
static int32_t settle_view_patches(kt_engine* e, hipStream_t s) {
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

// ---------------------------------------------------------------------------------------------------
// reconcile
// ---------------------------------------------------------------------------------------------------

// every aggregate launch gets an epoch; a workgroup stamps the slabs it spills with it (kt_reduce_bitmap_slabs then
// leaves alone what a namespace-ordered scan did not write)
static int32_t slab_tags(kt_engine* e, kt::AggScan& sc, hipStream_t s) {
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
static int32_t request_sums_in_range(kt_engine* e, hipStream_t s) {
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

static int32_t aggregate_locked(kt_engine* e, hipStream_t s, bool allow_fused = false) {
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
    if (!e->sw[kSw_NO_SCAN_VIEW]) {
      // scan-ordered copies of the listed pods' records: the scan streams them instead of gathering through the list
      // (namespace order for a multi-chunk index, ascending rows otherwise)
      // room for the pods that become countable before the next rebuild (kt_patch_scan_views appends them)
      const int64_t headroom = std::min<int64_t>(std::max<int64_t>(65536, (int64_t)e->n_countable / 16), e->cfg.pod_capacity - (int64_t)e->n_countable);
      e->view_cap_c = (int64_t)e->n_countable + headroom;
      e->view_extra = 0;
      const size_t nc = (size_t)e->view_cap_c + 1;
      // packed fold (PackPlan, kt_index.h) when every request of this engine is non-negative and the fields fit: sized
      // for the pods ONE workgroup scans with one workgroup per CU (two per CU scan fewer)
      if (!e->incremental && !e->wide && !e->sw[kSw_NO_PACK] && !e->dindex.has_long) {
        uint64_t slab_pods = kt::aggregate_slab_pods(e->view_cap_c, kt::aggregate_blocks(e->view_cap_c));
        // (planned ranges hold up to wg_range_cap records)
        if (e->range_c_G) slab_pods = std::max<uint64_t>(slab_pods, (uint64_t)kt::wg_range_cap((int64_t)e->n_countable, e->range_c_G) + 64u);
        e->pack = kt::make_pack_plan(e->D, e->max_abs, e->or_abs, e->neg_seen, slab_pods, /*pad_odd=*/true);
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
static int32_t delta_scan(kt_engine* e, int64_t n, const int64_t* rows_dev, int64_t row0, int sign, hipStream_t s) {
  if (!e->incremental || !e->agg_valid || n <= 0 || e->thr_rows_hi == 0) return KT_OK;
  kt::AggScan sc;
  sc.n = n, sc.rows = rows_dev, sc.row0 = row0, sc.counts = true, sc.sign = sign, sc.overflow_pods = e->n_overflow != 0;
  int32_t rc = slab_tags(e, sc, s);
  if (rc != KT_OK) return rc;
  const char* k = kt::launch_aggregate_indexed(e->pods, sc, e->sp, e->d_sp.p, e->dindex, e->d_agg.p, e->d_slab.p, s, nullptr);
  if (!k) return e->fail(KT_ERR_UNSUPPORTED, "a chunk of the selector index exceeds the aggregate kernel's LDS budget");
  KT_HIP(e, hipGetLastError());
  return KT_OK;
}

// consume: kt_reconcile_launch — nobody reads the partials after this finalize, which leaves them zeroed for the next scan
static int32_t finalize_locked(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, hipStream_t s, bool consume = false,
                               const uint8_t* row_mask = nullptr) {
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
                                               small ? &sm : nullptr, e->n_overflow != 0, by_ns ? &view : nullptr);
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
  kt::PackPlan plan = kt::make_pack_plan(e->D, e->max_abs, e->or_abs, e->neg_seen, kt::aggregate_slab_pods(n, nb), /*pad_odd=*/true);
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
  for (int64_t i = 0; i < n; ++i)
    for (int32_t j = 0; j < m; ++j) {
      const uint8_t v = st[(size_t)i * (size_t)T + (size_t)throttle_rows[j]];
      out[(size_t)i * (size_t)m + (size_t)j] = v == KT_STATUS_ERROR ? (uint8_t)KT_STATUS_ERROR : v != KT_STATUS_NOT_AFFECTED ? 1 : 0;
    }
  return KT_OK;
}

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
  int32_t rc;
  for (int32_t k = 0; k < n_pages; ++k)  // (enqueued on every page's own stream: the pages run side by side on the device)
    if ((rc = kt_reconcile_launch(pages[k], now_s, now_ns, flags, nullptr)) != KT_OK) return rc;
  if (replaced_any) memset(replaced_any, 0, (size_t)n);
  if (error_any) memset(error_any, 0, (size_t)n);
  for (int32_t k = 0; k < n_pages; ++k) {
    if ((rc = kt_reconcile_fetch(pages[k], n, &page_out[k])) != KT_OK) return rc;
    for (int32_t i = 0; i < n; ++i) {
      if (replaced_any && page_out[k].calc_at_nonzero) replaced_any[i] |= page_out[k].calc_at_nonzero[i] != 0;
      if (error_any && page_out[k].error) error_any[i] |= page_out[k].error[i] != 0;
    }
  }
  return KT_OK;
}

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

}  // extern "C"
