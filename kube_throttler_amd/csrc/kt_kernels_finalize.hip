// kt_kernels_finalize.hip — hand-written HIP kernels for gfx950 (CDNA4, wave64), throttle side: kt_finalize (used, CalculateThreshold,
// throttled flags, next override instant, CheckRecs), kt_reduce_finalize_packed (slab reduction + finalize as one launch),
// kt_prepare_check, and the DENSE pod x throttle scans in the reference's loop shape (cross-check variant).
//
// Everything here is integer / compare work on row tables in HBM: no MFMA, no floating point.
#include "kt_index_device.h"

namespace kt {

constexpr int kBlock = 256;

static inline int grid_for(int64_t n, int per_block = kBlock, int max_blocks = 256 * 8) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// ---------------------------------------------------------------------------------------------------
// Dense selector walk for one (pod lane, throttle t): terms in order, first match wins, an
// unconvertible podSelector reached before a match is an error (throttle_selector.go:30-54,
// clusterthrottle_selector.go:44-87).  t is wave-uniform => program loads are scalar.
// ---------------------------------------------------------------------------------------------------
template <int LT, bool KEYS>
__device__ __forceinline__ void walk_terms(const SelProgram& sp, int t, const uint32_t* ns_row, bool lane_on,
                                           const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], uint32_t& cur_w,
                                           uint32_t& cur_wi, bool& matched, bool& err) {
  matched = false;
  err = false;
  bool open = lane_on;
  const uint32_t g1 = sp.thr_term_off[t + 1];
  for (uint32_t g = sp.thr_term_off[t]; g < g1; ++g) {
    if ((g >> 5) != cur_wi) {
      cur_wi = g >> 5;
      cur_w = lane_on ? ns_row[cur_wi] : 0u;
    }
    const bool applies = open && ((cur_w >> (g & 31)) & 1u);
    if (!__any(applies)) continue;  // wave-uniform skip: no lane's namespace admits this term
    if (sp.term_flags[g] & kTermPodSelInvalid) {
      err |= applies;
      open &= !applies;
      continue;
    }
    const bool m = term_match<LT, KEYS>(sp, g, lp, lk) && applies;
    matched |= m;
    open &= !m;
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_aggregate_dense — affectedPods + fold ResourceAmount.Add (throttle_controller.go:116-119,221-246;
// clusterthrottle_controller.go:119-122,224-270; resource_amount.go:91-110) for all throttles.
// lane = pod, throttles walked uniformly; matched lanes add their request vector, key-presence
// counts and a pod count into partial[t][2D+2] (int64 sums: associative => any order, any #GPUs).
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlock) void kt_aggregate_dense(PodTable pods, int64_t n_rows, SelProgram sp,
                                                            unsigned long long* partial, int limb) {
  const int D = pods.D, stride = partial_stride(D);
  const int64_t n_round = (n_rows + kWave - 1) / kWave * kWave;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_round; p += (int64_t)gridDim.x * kBlock) {
    const bool in = p < n_rows;
    const uint32_t fl = in ? pods.flags[p] : 0u;
    // shouldCountIn: schedulerName == target && nodeName != "" (throttle_controller.go:217-219)
    const bool countable = (fl & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
    if (!__any(countable)) continue;
    PodRegs<DT, LT, KEYS> r;
    if (countable) load_pod<DT, LT, KEYS>(pods, p, r, true);
    else {
      r.ns = 0;
#pragma unroll
      for (int l = 0; l < LT; ++l) r.lp[l] = 0, r.lk[l] = 0;
#pragma unroll
      for (int d = 0; d < DT; ++d) r.v[d] = 0;
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) r.v[d] = limb_of(r.v[d], limb);
    const uint32_t present = fl >> kPresentShift;
    const bool not_finished = !(fl & kPodFinished);  // isNotFinished (pod_util.go:26-28)
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)r.ns * sp.gw;
    uint32_t cur_w = 0, cur_wi = 0xFFFFFFFFu;
    for (int t = 0; t < sp.T; ++t) {
      bool matched, err;
      walk_terms<LT, KEYS>(sp, t, ns_row, countable, r.lp, r.lk, cur_w, cur_wi, matched, err);
      unsigned long long* row = partial + (size_t)t * stride;
      if (err) atomicAdd(row + partial_off_errors(D), 1ull);
      if (matched && not_finished) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
          if (d < D && ((present >> d) & 1u)) {
            if (r.v[d] != 0) atomicAdd(row + d, (unsigned long long)r.v[d]);
            atomicAdd(row + partial_off_presence(D) + d, 1ull);
          }
        atomicAdd(row + partial_off_pods(D), 1ull);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_finalize — per throttle: used from the (all-reduced) partial sums, CalculateThreshold(now)
// (throttle_types.go:65-106, temporary_threshold_override.go:57-70), replace-only-if-changed
// (throttle_controller.go:122-132, Semantic.DeepEqual by value) and
// throttled = calculatedThreshold.IsThrottled(used, true) (:133, resource_amount.go:127-159).
//
// lane = (throttle, DIMENSION): a throttle is a group of DT consecutive lanes (8 throttles per wave at DT = 8), every
// lane holds ONE dimension of every amount.  Rounds 1-2 ran one thread per throttle: every load and store of a wave then
// touched 64 cache lines, and the D-unrolled logic was a 5 000-instruction serial stream per wave (13 us for 1000
// throttles, all of it latency).  Here a group's lanes read one 64-byte row together, the per-dimension work (used,
// first-wins override merge, changed-by-value, flags, the 128-bit headroom of the CheckRec) is one lane's scalar work,
// and every per-throttle BITMASK (presence, flags, active / tight dimensions) is the group's slice of a wave ballot.
// Per-throttle scalars (counts, flags, message fingerprints) are loaded by every lane of the group (one address: a
// broadcast) and stored by the lane of dimension 0.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int instant_cmp(int64_t as, int32_t an, int64_t bs, int32_t bn) {
  return as != bs ? (as < bs ? -1 : 1) : (an != bn ? (an < bn ? -1 : 1) : 0);
}

__device__ __forceinline__ bool cmp_eq(int64_t a, int64_t b, bool eq) { return eq ? a >= b : a > b; }
// used + reserved in 128 bits: the all-reduced `used` of several ranks may come close to int64's end
__device__ __forceinline__ bool cmp_eq_sum(__int128 a0, int64_t a1, int64_t b, bool eq) {
  const __int128 a = a0 + (__int128)a1;
  return eq ? a >= (__int128)b : a > (__int128)b;
}
// a `used` value from its two words (hi_valid = false: in int64 range, the sign extension of lo)
__device__ __forceinline__ __int128 wide_value(int64_t lo, int64_t hi, bool hi_valid) {
  return hi_valid ? (__int128)(((unsigned __int128)(uint64_t)hi << 64) | (unsigned __int128)(uint64_t)lo) : (__int128)lo;
}

// the DT-bit slice of a wave ballot that belongs to this lane's throttle: bit d = the predicate of dimension d.
// Control flow is uniform within a group (all its lanes share the throttle), so a ballot taken inside a branch still
// carries every lane of the groups that took it.
template <int DT>
__device__ __forceinline__ uint32_t group_bits(bool b) {
  const uint32_t lane = threadIdx.x & 63u;
  return (uint32_t)(__ballot(b) >> (lane & ~(uint32_t)(DT - 1))) & ((1u << DT) - 1u);
}

// Everything CheckThrottledFor needs that does not depend on the pod, folded into the throttle's CheckRec
// (effective threshold, headroom, step-2/3 bitmask, count verdicts) — see DESIGN.md "Check algebra".  This lane's
// dimension d: th_v / u_v / r_v are ITS values, the masks and counts are the throttle's.
// fl: the throttle's flags as stored AFTER this point (kThrCalcAtNonzero already decided the threshold passed in).
template <int DT>
__device__ __forceinline__ void build_check_rec(int t, int T, int D, int d, bool valid, uint32_t fl, int64_t th_v, uint32_t th_p, bool th_hc,
                                                int64_t th_c, __int128 u_v, uint32_t u_p, bool u_hc, int64_t u_c_, int64_t r_v, uint32_t r_p,
                                                bool r_hc, int64_t r_c_, uint32_t thrl_flag, uint32_t thrl_has, bool eq, const ReqBound& vmax,
                                                CheckRec<DT>* recs) {
  const int64_t u_c = u_hc ? u_c_ : 0, r_c = r_hc ? r_c_ : 0;
  const bool eq3 = (fl & kThrCluster) ? eq : true;  // throttle_types.go:143 vs clusterthrottle_types.go:45
  uint32_t f = 0;
  // step 1, counts: IsThrottled(podAmount{pod:1}, false).pod
  if (th_hc && 1 > th_c) f |= kRecExceedsByCount;
  // step 2: stored status.throttled ; step 3: IsThrottled(used + reserved, eq3)
  bool act_pod = (fl & kThrThrottledPod) != 0;
  if (th_hc && (u_hc || r_hc) && cmp_eq_sum(u_c, r_c, th_c, eq3)) act_pod = true;
  // step 4, counts: used + pod(1) + reserved always has counts
  if (th_hc && (eq ? (__int128)u_c + 1 + r_c >= (__int128)th_c : (__int128)u_c + 1 + r_c > (__int128)th_c)) f |= kRecInsufficientByCount;
  int64_t thr = kInf, head = kInf;
  bool act_d = false;
  const bool in_d = valid && d < D;
  if (in_d && ((th_p >> d) & 1u)) {
    const __int128 uv = ((u_p >> d) & 1u) ? u_v : (__int128)0;
    const int64_t rv = ((r_p >> d) & 1u) ? r_v : 0;
    act_d = (((u_p | r_p) >> d) & 1u) && cmp_eq_sum(uv, rv, th_v, eq3);
    thr = th_v;
    const __int128 h = (__int128)th_v - uv - (__int128)rv - (eq ? 1 : 0);  // |uv| < 2^125: no 128-bit overflow
    head = h >= (__int128)INT64_MAX ? kInf : h <= (__int128)INT64_MIN ? INT64_MIN : (int64_t)h;
  }
  const uint32_t act_mask = (thrl_flag & thrl_has) | group_bits<DT>(act_d);
  // could ANY pod of this engine exceed the threshold in this dimension — or the headroom, where that can still
  // change the verdict (a pod that requests an already-active dimension is `active` whatever the headroom says)
  const int64_t vm = vmax.v[d < 16 ? d : 15];
  const bool tight_d = in_d && (vm > thr || (!act_pod && !((act_mask >> d) & 1u) && vm > head));
  if (group_bits<DT>(tight_d) != 0u) f |= kRecTight;
  if (act_pod) f |= kRecActiveByCount;
  if (!valid) return;
  recs[t].thr[d] = thr;
  recs[t].head[d] = head;
  if (d == 0) {
    recs[t].flags = f;
    recs[t].active_mask = act_mask;
    rec_flags<DT>(recs, T)[t] = RecFlags{f, act_mask};
  }
}

// One throttle's stored state as the lane of dimension d sees it, requested as ONE batch of independent loads (the
// kernel is a latency chain: every load that waits for a branch outcome is another round trip).
struct ThrLane {
  uint32_t fl, thrl_flag, thrl_has, ovr0, ovr1;
  uint64_t status_fp, spec_fp;
  int64_t calc_v, used_v, spec_v, res_v;  // this lane's dimension; 0 for padding dimensions
  int64_t used_hi_raw;                    // tt.used_hi's word (0 when the table is absent: hi() then extends the sign)
  bool has_hi;
  // high word of used_v.  Computed at the point of use: nothing between the loads may wait for one of them
  __device__ __forceinline__ int64_t hi() const { return has_hi ? used_hi_raw : (used_v < 0 ? (int64_t)-1 : (int64_t)0); }
  uint32_t calc_p, used_p, spec_p, res_p;
  int64_t calc_c, used_c, spec_c, res_c;
  bool calc_hc, used_hc, spec_hc, res_hc;
};
// the loads alone: nothing here looks at a loaded value, so a caller can put further independent loads behind them
// before anything waits (padding dimensions do not load: their values stay 0)
struct ThrRaw {
  uint32_t fl, thrl_flag, thrl_has, ovr0, ovr1;
  uint64_t status_fp, spec_fp;
  int64_t calc_v, used_v, spec_v, res_v, used_hi;
  uint32_t calc_p, used_p, spec_p, res_p;
  int64_t calc_c, used_c, spec_c, res_c;
  uint8_t calc_hc, used_hc, spec_hc, res_hc;
  bool has_hi;
};
__device__ __forceinline__ void load_thr_raw(const ThrTables& tt, int t, int D, int d, ThrRaw& r) {
  r.fl = tt.flags[t], r.thrl_flag = tt.thrl_flag[t], r.thrl_has = tt.thrl_has[t];
  r.ovr0 = tt.ovr_off[t], r.ovr1 = tt.ovr_off[t + 1];
  r.status_fp = tt.status_msgs_fp[t], r.spec_fp = tt.spec_msgs_fp[t];
  r.calc_p = tt.calc.present[t], r.used_p = tt.used.present[t], r.spec_p = tt.spec.present[t], r.res_p = tt.reserved.present[t];
  r.calc_c = tt.calc.count[t], r.used_c = tt.used.count[t], r.spec_c = tt.spec.count[t], r.res_c = tt.reserved.count[t];
  r.calc_hc = tt.calc.has_count[t], r.used_hc = tt.used.has_count[t], r.spec_hc = tt.spec.has_count[t], r.res_hc = tt.reserved.has_count[t];
  r.calc_v = r.used_v = r.spec_v = r.res_v = r.used_hi = 0;
  r.has_hi = tt.used_hi != nullptr;
  if (d < D) {
    const size_t i = (size_t)t * D + d;
    r.calc_v = tt.calc.v[i], r.used_v = tt.used.v[i], r.spec_v = tt.spec.v[i], r.res_v = tt.reserved.v[i];
    if (tt.used_hi) r.used_hi = tt.used_hi[i];
  }
}
__device__ __forceinline__ void thr_from_raw(const ThrRaw& w, ThrLane& r) {
  r.fl = w.fl, r.thrl_flag = w.thrl_flag, r.thrl_has = w.thrl_has, r.ovr0 = w.ovr0, r.ovr1 = w.ovr1;
  r.status_fp = w.status_fp, r.spec_fp = w.spec_fp;
  r.calc_v = w.calc_v, r.used_v = w.used_v, r.spec_v = w.spec_v, r.res_v = w.res_v, r.used_hi_raw = w.used_hi, r.has_hi = w.has_hi;
  r.calc_p = w.calc_p, r.used_p = w.used_p, r.spec_p = w.spec_p, r.res_p = w.res_p;
  r.calc_c = w.calc_c, r.used_c = w.used_c, r.spec_c = w.spec_c, r.res_c = w.res_c;
  r.calc_hc = w.calc_hc != 0, r.used_hc = w.used_hc != 0, r.spec_hc = w.spec_hc != 0, r.res_hc = w.res_hc != 0;
}
__device__ __forceinline__ void load_thr(const ThrTables& tt, int t, int D, int d, ThrLane& r) {
  ThrRaw w;
  load_thr_raw(tt, t, D, d, w);
  thr_from_raw(w, r);
}

// the CheckRec of throttle t from the status as held in the lane registers (stored status unchanged)
template <int DT>
__device__ __forceinline__ void build_check_rec_regs(int t, int T, int D, int d, bool valid, const ThrLane& r, bool eq, const ReqBound& vmax,
                                                     CheckRec<DT>* recs) {
  // threshold := status.calculatedThreshold if calculatedAt != zero else spec.threshold (throttle_types.go:129-132)
  const bool calc = (r.fl & kThrCalcAtNonzero) != 0;
  build_check_rec<DT>(t, T, D, d, valid, r.fl, calc ? r.calc_v : r.spec_v, calc ? r.calc_p : r.spec_p, calc ? r.calc_hc : r.spec_hc,
                      calc ? r.calc_c : r.spec_c, wide_value(r.used_v, r.hi(), true), r.used_p, r.used_hc, r.used_c, r.res_v, r.res_p, r.res_hc, r.res_c, r.thrl_flag,
                      r.thrl_has, eq, vmax, recs);
}

// One throttle of kt_finalize, the lane of dimension d.  pv / pc: its words of the partial row (value, per-key
// contributor count); pods / errs: the row's pod count and error count.
template <int DT>
__device__ __forceinline__ void finalize_throttle(const ThrTables& tt, int t, int T, int D, int d, bool valid, const ThrLane& r,
                                                  unsigned long long pv, unsigned long long pc, unsigned long long pods,
                                                  unsigned long long errs, int64_t now_s, int32_t now_ns, int apply, const ReconcileOut& out,
                                                  CheckRec<DT>* recs, int rec_eq, const ReqBound& vmax, bool selected, bool wide = false,
                                                  unsigned long long pv_hi = 0) {
  // wide: the sums came as two limb sums — pv = sum of the low 32 bits of every request, pv_hi = sum of request >> 32
  // (arithmetic) — because their total leaves int64 (resource.Quantity would have promoted, resourcelist.go:48-54)
  // recs (nullable): also leave the CheckRec of the throttle for the check that follows (kt_prepare_check fused in:
  // saves one dependent launch per reconcile + check step); rec_eq = the isThrottledOnEqual value it is built for
  const uint32_t fl = r.fl;
  const bool in_d = valid && d < D, lead = valid && d == 0;
  const size_t vi = (size_t)t * D + (d < D ? d : 0);
  // selected = false: a reconcile of other keys (kt_reconcile_rows_launch) — this throttle keeps its stored status
  const bool live = selected && (fl & (kThrValid | kThrResponsible)) == (kThrValid | kThrResponsible);
  const bool error = live && errs != 0;
  if (!live || error) {  // the stored status is returned unchanged
    if (in_d) {
      out.used.v[vi] = r.used_v;
      if (out.used_hi) out.used_hi[vi] = r.hi();
      out.calc.v[vi] = r.calc_v;
    }
    if (lead) {
      out.used.present[t] = r.used_p;
      out.used.count[t] = r.used_c;
      out.used.has_count[t] = r.used_hc;
      out.calc.present[t] = r.calc_p;
      out.calc.count[t] = r.calc_c;
      out.calc.has_count[t] = r.calc_hc;
      out.calc_updated[t] = 0;
      out.thrl_flag[t] = r.thrl_flag;
      out.thrl_has[t] = r.thrl_has;
      out.thrl_pod[t] = (fl & kThrThrottledPod) ? 1 : 0;
      out.error[t] = error ? 1 : 0;
      out.next_s[t] = INT64_MAX;  // reconcile returns before NextOverrideHappensIn (throttle_controller.go:103-111)
      out.next_ns[t] = 0;
    }
    if (recs) build_check_rec_regs<DT>(t, T, D, d, valid, r, rec_eq != 0, vmax, recs);
    return;
  }
  // ---- used = fold Add over counted pods (zero matches => ResourceAmount{}: counts nil, requests nil)
  // a key is present when some counted pod carried it: the contributor count says so, and so does a non-zero sum
  const bool u_pr = in_d && (pc != 0 || pv != 0 || pv_hi != 0);
  const uint32_t u_p = group_bits<DT>(u_pr);
  const __int128 u_w = !u_pr ? (__int128)0 : wide ? (__int128)(unsigned __int128)pv + (__int128)(int64_t)pv_hi * ((__int128)1 << 32) : (__int128)(int64_t)pv;
  const int64_t u_v = (int64_t)(uint64_t)u_w, u_hi = (int64_t)(u_w >> 64);
  const int64_t u_c = (int64_t)pods;
  const bool u_hc = u_c > 0;
  // ---- CalculateThreshold(now)
  int64_t c_v = 0;
  bool c_pd = false;  // this dimension of the merged override
  bool c_hc = false, active_found = false, any_err = false;
  int64_t c_c = 0;
  // NextOverrideHappensIn (throttle_types.go:37-63): earliest begin / end instant strictly after now
  int64_t nx_s = INT64_MAX;
  int32_t nx_ns = 0;
  auto sooner = [&](int64_t s_, int32_t ns_) {
    if (instant_cmp(s_, ns_, now_s, now_ns) > 0 && instant_cmp(s_, ns_, nx_s, nx_ns) < 0) nx_s = s_, nx_ns = ns_;
  };
  for (uint32_t o = r.ovr0; o < r.ovr1; ++o) {
    // one batch of loads per override
    const uint8_t of = tt.ovr_flags[o];
    const int64_t ob_s = tt.ovr_begin_s[o], oe_s = tt.ovr_end_s[o];
    const int32_t ob_ns = tt.ovr_begin_ns[o], oe_ns = tt.ovr_end_ns[o];
    const bool o_hc = tt.ovr_thr.has_count[o] != 0;
    const int64_t o_c = tt.ovr_thr.count[o];
    const uint32_t op = tt.ovr_thr.present[o];
    const int64_t o_v = tt.ovr_thr.v[(size_t)o * D + (d < D ? d : 0)];
    if (of & kOvrParseError) {
      any_err = true;
      if (of & kOvrBeginParsed) sooner(ob_s, ob_ns);  // only `end` is bad
      continue;
    }
    sooner(ob_s, ob_ns);
    sooner(oe_s, oe_ns);
    const bool begin = instant_cmp(ob_s, ob_ns, now_s, now_ns) <= 0;
    const bool end_zero = oe_s == kZeroTimeS && oe_ns == 0;
    const bool end = end_zero || instant_cmp(now_s, now_ns, oe_s, oe_ns) <= 0;
    if (!(begin && end)) continue;
    active_found = true;
    if (!c_hc && o_hc) {  // first active override wins, per resource and for counts
      c_hc = true;
      c_c = o_c;
    }
    if (in_d && ((op >> d) & 1u) && !c_pd) {
      c_pd = true;
      c_v = o_v;
    }
  }
  if (!active_found) {  // no active override: spec.threshold; otherwise the merged override REPLACES it
    c_pd = in_d && ((r.spec_p >> d) & 1u);
    c_hc = r.spec_hc;
    c_c = r.spec_c;
    c_v = c_pd ? r.spec_v : 0;
  }
  uint32_t c_p = group_bits<DT>(c_pd);
  const uint64_t c_fp = any_err ? r.spec_fp : 0ull;
  // ---- replace the stored calculatedThreshold only if threshold or messages differ by value
  const bool differs_d = c_pd && c_v != r.calc_v;
  const bool same = (c_hc == r.calc_hc) && (!c_hc || c_c == r.calc_c) && (c_p == r.calc_p) && group_bits<DT>(differs_d) == 0u;
  const bool replace = !same || r.status_fp != c_fp;
  if (!replace) {
    c_p = r.calc_p;
    c_hc = r.calc_hc;
    c_c = r.calc_c;
    c_v = r.calc_v;
    c_pd = in_d && ((c_p >> d) & 1u);
  }
  // ---- throttled = calculatedThreshold.IsThrottled(used, onEqual = true)
  const bool th_pod = c_hc && u_hc && u_c >= c_c;
  const uint32_t th_flag = group_bits<DT>(c_pd && u_pr && u_w >= (__int128)c_v);
  // ---- outputs
  const int64_t c_out = c_pd ? c_v : 0;
  if (in_d) {
    out.used.v[vi] = u_v;
    if (out.used_hi) out.used_hi[vi] = u_hi;
    out.calc.v[vi] = c_out;
  }
  if (lead) {
    out.used.present[t] = u_p;
    out.used.count[t] = u_hc ? u_c : 0;
    out.used.has_count[t] = u_hc;
    out.calc.present[t] = c_p;
    out.calc.count[t] = c_hc ? c_c : 0;
    out.calc.has_count[t] = c_hc;
    out.calc_updated[t] = replace;
    out.thrl_flag[t] = th_flag;
    out.thrl_has[t] = c_p;
    out.thrl_pod[t] = th_pod;
    out.error[t] = 0;
    out.next_s[t] = nx_s;
    out.next_ns[t] = nx_ns;
  }
  if (apply) {  // UpdateStatus: the result becomes the stored status the next check reads
    if (in_d) {
      tt.used.v[vi] = u_v;
      if (tt.used_hi) tt.used_hi[vi] = u_hi;
      if (replace) tt.calc.v[vi] = c_out;
    }
    uint32_t nf = fl & ~kThrThrottledPod;
    if (th_pod) nf |= kThrThrottledPod;
    if (replace) nf |= kThrCalcAtNonzero;
    if (lead) {
      tt.used.present[t] = u_p;
      tt.used.count[t] = u_hc ? u_c : 0;
      tt.used.has_count[t] = u_hc;
      if (replace) {
        tt.calc.present[t] = c_p;
        tt.calc.count[t] = c_hc ? c_c : 0;
        tt.calc.has_count[t] = c_hc;
        tt.status_msgs_fp[t] = c_fp;
      }
      tt.flags[t] = nf;
      tt.thrl_flag[t] = th_flag;
      tt.thrl_has[t] = c_p;
    }
    if (recs) {  // from the registers that were just stored (no re-read of this lane's own writes)
      const bool calc = (nf & kThrCalcAtNonzero) != 0;  // calculatedAt still zero: spec.threshold
      build_check_rec<DT>(t, T, D, d, valid, nf, calc ? c_out : r.spec_v, calc ? c_p : r.spec_p, calc ? c_hc : r.spec_hc, calc ? c_c : r.spec_c,
                          u_w, u_p, u_hc, u_c, r.res_v, r.res_p, r.res_hc, r.res_c, th_flag, c_p, rec_eq != 0, vmax, recs);
    }
  } else if (recs) {
    build_check_rec_regs<DT>(t, T, D, d, valid, r, rec_eq != 0, vmax, recs);
  }
}

constexpr int kFinalizeBlock = 64;  // one wave: 64 / DT throttles
// partial_hi (nullable): wide sums — `partial` then holds the sums of the requests' low 32-bit limbs and partial_hi (same
// layout; only its value words matter) the sums of their high parts
template <int DT>
__global__ __launch_bounds__(kFinalizeBlock) void kt_finalize(ThrTables tt, int T, int D, unsigned long long* partial, unsigned long long* partial_hi,
                                                                int consume, int64_t now_s, int32_t now_ns, int apply, ReconcileOut out,
                                                                CheckRec<DT>* recs, int rec_eq, const ReqBound vmax, const uint8_t* row_mask) {
  // consume: leave the row zeroed behind (kt_reconcile_launch: the next aggregate then needs no clearing pass)
  const int d = (int)(threadIdx.x & (DT - 1));
  const int tq = (int)((blockIdx.x * kFinalizeBlock + threadIdx.x) / DT);
  const bool valid = tq < T;
  const int t = valid ? tq : T - 1;  // lanes past the end shadow the last throttle (every lane takes part in the ballots) and store nothing
  const int stride = partial_stride(D);
  ThrLane r;
  load_thr(tt, t, D, d, r);
  unsigned long long* prow = partial + (size_t)t * stride;
  const int dd = d < D ? d : 0;
  const unsigned long long a = prow[dd], b = prow[partial_off_presence(D) + dd];
  const unsigned long long pv = d < D ? a : 0ull, pc = d < D ? b : 0ull;
  const unsigned long long pods = prow[partial_off_pods(D)], errs = prow[partial_off_errors(D)];
  unsigned long long pv_hi = 0;
  if (partial_hi) {
    const unsigned long long h = partial_hi[(size_t)t * stride + dd];
    pv_hi = d < D ? h : 0ull;
  }
  if (consume && valid) {
    // the group's lanes clear the row between them: words d, d + DT, ... (every word was read above by some lane of the
    // group before any lane of it stores: the loads complete before dependent code, the stores follow the ballots below)
    for (int j = d; j < stride; j += DT) prow[j] = 0ull;
    if (partial_hi)
      for (int j = d; j < stride; j += DT) partial_hi[(size_t)t * stride + j] = 0ull;
  }
  finalize_throttle<DT>(tt, t, T, D, d, valid, r, pv, pc, pods, errs, now_s, now_ns, apply, out, recs, rec_eq, vmax,
                        row_mask == nullptr || row_mask[t] != 0, partial_hi != nullptr, pv_hi);
}

void launch_finalize(const ThrTables& tt, const SelProgram& sp, int D, unsigned long long* partial, bool consume,
                     int64_t now_s, int32_t now_ns, bool apply, const ReconcileOut& out, void* recs, int rec_DT, bool rec_eq,
                     const ReqBound& vmax, hipStream_t s, const uint8_t* row_mask, unsigned long long* partial_hi) {
  if (sp.T <= 0) return;
  const int DT = recs ? rec_DT : (D <= 4 ? 4 : D <= 8 ? 8 : 16);  // the CheckRec layout follows the check kernel
  // one wave per 64 / DT throttles (T is small: spread over as many CUs as possible; everything is latency)
  const dim3 g((unsigned)(((size_t)sp.T * DT + kFinalizeBlock - 1) / kFinalizeBlock)), b(kFinalizeBlock);
  const int eq = rec_eq ? 1 : 0;
  if (DT == 4) hipLaunchKernelGGL(kt_finalize<4>, g, b, 0, s, tt, sp.T, D, partial, partial_hi, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<4>*)recs, eq, vmax, row_mask);
  else if (DT == 8) hipLaunchKernelGGL(kt_finalize<8>, g, b, 0, s, tt, sp.T, D, partial, partial_hi, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<8>*)recs, eq, vmax, row_mask);
  else hipLaunchKernelGGL(kt_finalize<16>, g, b, 0, s, tt, sp.T, D, partial, partial_hi, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<16>*)recs, eq, vmax, row_mask);
}

// kt_reduce_finalize_packed — the slab reduction of a packed scan (kt_aggregate_bitmap PK) and kt_finalize as ONE
// launch.  A block of 16 waves takes a tile of whole records (64 / units of them) of one chunk's slab row:
//   * the threads that will finalize — thread = (record of the tile, dimension), 64 / DT throttles per wave exactly as
//     in kt_finalize — first request their record's throttle and its stored state (two dependent batches of loads);
//   * all 16 waves sum the tile over the workgroups' slabs (block_record_sums, kt_index_device.h: coalesced, one batch
//     of loads in flight beside the ones above) into LDS;
//   * the finalizing threads cut their totals out of the LDS sums and — the record being the throttle's only group — go
//     straight on to finalize_throttle.  Two dependent launches (a reduction that funnels 10 MB into 140 KB through
//     atomics, then a latency chain over the tables) become one, and the sums never travel through the partial buffer.
//   * a throttle with several groups (namespace cells): every group adds its sums to the throttle's partial row and takes
//     a ticket; the last to arrive finalizes from the row (and leaves row and ticket zeroed);
//   * what the scan kernel itself added to the partial buffer (slow-list throttles, overflow pods, selector errors) is
//     read from the row and added; throttles without any group get blocks of their own (blockIdx.y = chunks).
// kt_reconcile_launch only: with several ranks the partial rows cross the all-reduce between reduction and kt_finalize
// (kt_aggregate_launch -> exchange -> kt_finalize_launch keep the separate kernels).
struct FusedReduceArgs {
  const unsigned char* slab;
  const BmChunk* chunks;
  const uint32_t* rank_t;
  const uint32_t* thr_ngrp;
  const uint32_t* nogroup;
  uint32_t* arrive;
  const uint32_t* slab_tag;
  uint32_t n_chunks, n_nogroup, epoch;
  int32_t n_slabs, check_tags;
  int32_t scan_adds_rows;  // the scan kernel itself may have added to partial rows (slow-list throttles, overflow pods)
  PackPlan pk;
  BmChunk ch0;             // n_chunks == 1: the chunk's descriptor by value (one dependent load less)
};
template <int DT>
__global__ __launch_bounds__(kRecBlock) void kt_reduce_finalize_packed(const FusedReduceArgs f, ThrTables tt, int T, int D, unsigned long long* partial,
                                                                      int consume, int64_t now_s, int32_t now_ns, int apply, ReconcileOut out,
                                                                      CheckRec<DT>* recs, int rec_eq, const ReqBound vmax, const uint8_t* row_mask) {
  __shared__ RecSumsLds lds;
  const uint32_t x = threadIdx.x;
  const int d = (int)(x & (DT - 1));
  const uint32_t g = x / DT;  // this thread's throttle of the block
  const int stride = partial_stride(D);
  unsigned long long pv = 0, pc = 0, pods = 0, errs = 0;
  uint32_t t;
  bool valid;                              // stores something (lanes past the end shadow the last throttle: ballots)
  bool from_row = f.scan_adds_rows != 0;  // (part of) the sums are in the throttle's partial row ...
  bool met = false;                       // ... put there by other groups of this launch
  ThrLane r;
  if (blockIdx.y < f.n_chunks) {
    BmChunk ch = f.ch0;
    if (f.n_chunks != 1u) {  // the descriptor as scalars, here and now (nothing later may wait for it behind other loads)
      const BmChunk* cp = f.chunks + blockIdx.y;
      ch.n_thr = __builtin_amdgcn_readfirstlane(cp->n_thr), ch.rank0 = __builtin_amdgcn_readfirstlane(cp->rank0);
      ch.slab_off = __builtin_amdgcn_readfirstlane(cp->slab_off);
    }
    const uint32_t units = f.pk.rec_bytes >> 3, rb = (uint32_t)kRecTileUnits / units;
    const uint32_t rec0 = blockIdx.x * rb;
    if (rec0 >= ch.n_thr) return;  // block-uniform
    const uint32_t nrec = min(rb, ch.n_thr - rec0);
    const bool fin = (x & ~63u) < rb * DT;  // wave-uniform: this wave finalizes
    valid = g < nrec;
    const uint32_t gl = valid ? g : nrec - 1u;
    // The order of the loads is the design (vmcnt counts in order: waiting for a load waits for every load before it):
    // the slab tags (multi-chunk programs only), the record's throttle, the sixteen slab words — then, the throttle
    // known, its stored state; the slab words are summed and exchanged through LDS while that last batch is in flight.
    const size_t pitch = ((size_t)ch.n_thr * f.pk.rec_bytes + 15u) & ~(size_t)15u;
    RecSlabLoads sl;
    record_slabs_live(f.n_slabs, f.slab_tag + blockIdx.y * kSlabTagStride, f.epoch, f.check_tags, sl);
    const unsigned char* row0 = f.slab + (size_t)ch.slab_off * 16 + (size_t)rec0 * f.pk.rec_bytes;
    if (!fin) {  // (its own copy of the code: the waits of the finalizing waves must not be planned for both kinds)
      record_slabs_issue(row0, pitch, nrec * units, sl);
      block_record_sums(sl, f.pk, lds);
      return;
    }
    t = f.rank_t[ch.rank0 + rec0 + gl];
    record_slabs_issue(row0, pitch, nrec * units, sl);
    ThrRaw raw;
    load_thr_raw(tt, (int)t, D, d, raw);
    uint32_t ngrp = f.thr_ngrp[t];
    block_record_sums(sl, f.pk, lds);
    // (the stored state is first looked at HERE: the compiler must not pull the byte -> bool conversions up into the loads)
    asm volatile("" : "+v"(raw.fl), "+v"(ngrp));
    thr_from_raw(raw, r);
    const uint32_t ub = gl * units;
    const unsigned long long rec_pods = packed_pods(lds, ub, f.pk);
    const uint32_t zero_keys = packed_zero_keys(lds, ub, f.pk);
    const unsigned long long mine = rec_pods ? packed_field(lds, ub, packed_desc_of(f.pk, d, D)) : 0ull;
    // ---- a throttle with several groups: they meet in the partial row, the last group to arrive goes on.  The lane of
    // dimension d adds that dimension (and its key mark), lane 0 the pod count too — RETURNING atomics: their results feed
    // the ticket's operand (through a ballot over the group), so the ticket is issued after every add has been performed at
    // the device's coherence point — ordering by data dependence instead of a release fence (an agent-scope fence writes
    // back and invalidates the XCD's whole L2: 20 000 waves doing that serialised the configs[4] launch into 0.9 ms)
    const bool multi = valid && ngrp > 1u;
    unsigned long long* prow = partial + (size_t)t * stride;
    unsigned long long seen = 0;
    if (multi && rec_pods) {
      if (d == 0) seen |= atomicAdd(prow + partial_off_pods(D), rec_pods);
      if (d < D) {
        if (mine) seen |= atomicAdd(prow + d, mine);
        if ((zero_keys >> d) & 1u) seen |= atomicAdd(prow + partial_off_presence(D) + d, 1ull);
      }
    }
    // (the previous values are sums far below 2^64: the comparison is false, but only the hardware knows)
    const uint32_t one = 1u + (group_bits<DT>(seen == ~0ull) != 0u ? 1u : 0u);
    uint32_t arrived = 0;
    if (multi && d == 0) arrived = __hip_atomic_fetch_add(f.arrive + t, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = (uint32_t)__shfl((int)arrived, (int)((x & 63u) & ~(uint32_t)(DT - 1)));  // the group's lane 0
    const bool last = multi && arrived + 1u == ngrp;
    if (last && d == 0) __hip_atomic_store(f.arrive + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (multi) {
      valid = last;  // the other groups are done: their lanes shadow on
      from_row = from_row || last, met = last;
    } else if (rec_pods) {
      pods = rec_pods;
      pv = d < D ? mine : 0ull;
      pc = d < D ? (zero_keys >> d) & 1u : 0u;  // a non-zero sum marks the key by itself
    }
  } else {
    const uint32_t i = blockIdx.x * (uint32_t)(kRecBlock / DT) + g;
    if ((i & ~(uint32_t)(64 / DT - 1)) >= f.n_nogroup) return;  // wave-uniform: no throttle left for this wave
    valid = i < f.n_nogroup;
    t = f.nogroup[valid ? i : f.n_nogroup - 1u];
    load_thr(tt, (int)t, D, d, r);
  }
  // ---- what the scan kernel added to the row by itself (slow list, overflow pods, errors) and, for a throttle of
  //      several groups, the sums of all its records
  {
    unsigned long long* prow = partial + (size_t)t * stride;
    // A row other groups of THIS launch added to is read by read-modify-writes — exchange with 0 (consume) or add 0 — at
    // the very point where those adds were performed: no cache level can answer with an older value, and the row is left
    // zeroed in the same operation.  The lane of dimension d takes words d and D + d, the group's first lane the pod
    // and error counts, handed to the other lanes below.
    unsigned long long p0 = 0, e0 = 0;
    if (from_row && met) {
      auto take = [&](int j) {
        return consume ? __hip_atomic_exchange(prow + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : __hip_atomic_fetch_add(prow + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      if (d < D) pv += take(d), pc += take(partial_off_presence(D) + d);
      if (d == 0) p0 = take(partial_off_pods(D)), e0 = take(partial_off_errors(D));
    }
    const int leader = (int)((x & 63u) & ~(uint32_t)(DT - 1));
    p0 = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)p0, leader) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(p0 >> 32), leader) << 32;
    e0 = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)e0, leader) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(e0 >> 32), leader) << 32;
    pods += p0, errs += e0;
    // rows only the scan kernel wrote (slow list, overflow pods, errors) are plain data of the previous launch
    if (from_row && !met) {
      const int dd = d < D ? d : 0;
      const unsigned long long a = prow[dd], b = prow[partial_off_presence(D) + dd], pp = prow[partial_off_pods(D)], ee = prow[partial_off_errors(D)];
      pv += d < D ? a : 0ull, pc += d < D ? b : 0ull, pods += pp, errs += ee;
      if (consume && valid)
        for (int j = d; j < stride; j += DT) prow[j] = 0ull;
    }
  }
  finalize_throttle<DT>(tt, (int)t, T, D, d, valid, r, pv, pc, pods, errs, now_s, now_ns, apply, out, recs, rec_eq, vmax,
                        row_mask == nullptr || row_mask[t] != 0);
}

void launch_reduce_finalize_packed(const ThrTables& tt, const SelProgram& sp, int D, const IndexDev& ix, const PackPlan& pk, const void* slab,
                                   int n_slabs, const uint32_t* slab_tag, uint32_t epoch, unsigned long long* partial, bool consume, int64_t now_s,
                                   int32_t now_ns, bool apply, const ReconcileOut& out, void* recs, int rec_DT, bool rec_eq, const ReqBound& vmax,
                                   hipStream_t s, const uint8_t* row_mask, bool scan_adds_rows) {
  if (sp.T <= 0) return;
  FusedReduceArgs f{};
  f.slab = (const unsigned char*)slab, f.chunks = ix.bm_chunks, f.rank_t = ix.bm_rank_t, f.thr_ngrp = ix.thr_ngrp, f.nogroup = ix.nogroup;
  f.arrive = ix.grp_arrive, f.slab_tag = slab_tag, f.n_chunks = ix.n_chunks, f.n_nogroup = ix.n_nogroup, f.epoch = epoch;
  f.n_slabs = n_slabs, f.check_tags = ix.n_chunks > 1 ? 1 : 0, f.pk = pk;
  f.scan_adds_rows = scan_adds_rows ? 1 : 0;
  if (ix.n_chunks == 1 && !ix.h_chunks.empty()) f.ch0 = ix.h_chunks[0];
  const int DT = recs ? rec_DT : (D <= 4 ? 4 : D <= 8 ? 8 : 16);
  const uint32_t rb = (uint32_t)kRecTileUnits / (pk.rec_bytes >> 3), per_ng = (uint32_t)(kRecBlock / DT);
  const uint32_t gx = std::max((ix.bm_max_thr + rb - 1u) / rb, (ix.n_nogroup + per_ng - 1u) / per_ng);
  const dim3 g(gx ? gx : 1u, ix.n_chunks + (ix.n_nogroup ? 1u : 0u)), b(kRecBlock);  // the extra row: throttles without a group
  const int eq = rec_eq ? 1 : 0;
  if (DT == 4) hipLaunchKernelGGL(kt_reduce_finalize_packed<4>, g, b, 0, s, f, tt, sp.T, D, partial, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<4>*)recs, eq, vmax, row_mask);
  else if (DT == 8) hipLaunchKernelGGL(kt_reduce_finalize_packed<8>, g, b, 0, s, f, tt, sp.T, D, partial, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<8>*)recs, eq, vmax, row_mask);
  else hipLaunchKernelGGL(kt_reduce_finalize_packed<16>, g, b, 0, s, f, tt, sp.T, D, partial, consume ? 1 : 0, now_s, now_ns, apply ? 1 : 0, out, (CheckRec<16>*)recs, eq, vmax, row_mask);
}

// ---------------------------------------------------------------------------------------------------
// kt_prepare_check — per throttle: fold everything CheckThrottledFor needs that does not depend on
// the pod into a CheckRec (effective threshold, headroom, step-2/3 bitmask, count verdicts); lane = (throttle,
// dimension) like kt_finalize.
// ---------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(kFinalizeBlock) void kt_prepare_check(ThrTables tt, int T, int D, int on_equal, CheckRec<DT>* recs, const ReqBound vmax) {
  const int d = (int)(threadIdx.x & (DT - 1));
  const int tq = (int)((blockIdx.x * kFinalizeBlock + threadIdx.x) / DT);
  const bool valid = tq < T;
  const int t = valid ? tq : T - 1;
  ThrLane r;
  load_thr(tt, t, D, d, r);
  build_check_rec_regs<DT>(t, T, D, d, valid, r, on_equal != 0, vmax, recs);
}

void launch_prepare_check(const ThrTables& tt, int T, int D, int DT, bool on_equal, void* recs, const ReqBound& vmax, hipStream_t s) {
  if (T <= 0) return;
  dim3 g((unsigned)(((size_t)T * DT + kFinalizeBlock - 1) / kFinalizeBlock)), b(kFinalizeBlock);
  if (DT == 4) hipLaunchKernelGGL(kt_prepare_check<4>, g, b, 0, s, tt, T, D, on_equal ? 1 : 0, (CheckRec<4>*)recs, vmax);
  else if (DT == 8) hipLaunchKernelGGL(kt_prepare_check<8>, g, b, 0, s, tt, T, D, on_equal ? 1 : 0, (CheckRec<8>*)recs, vmax);
  else hipLaunchKernelGGL(kt_prepare_check<16>, g, b, 0, s, tt, T, D, on_equal ? 1 : 0, (CheckRec<16>*)recs, vmax);
}

// ---------------------------------------------------------------------------------------------------
// kt_check_dense — PreFilter for n pods: CheckThrottled of both controllers + CheckThrottledFor
// (plugin.go:148-215, throttle_controller.go:349-397, clusterthrottle_controller.go:378-425).
// lane = pod; throttles walked uniformly (records and selector program come through scalar loads).
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlock) void kt_check_dense(PodTable pods, int64_t n, const int64_t* rows, SelProgram sp,
                                                        const void* recs_, uint64_t* summary, uint8_t* status) {
  const CheckRec<DT>* recs = (const CheckRec<DT>*)recs_;
  const int64_t n_round = (n + kWave - 1) / kWave * kWave;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * kBlock) {
    const bool in = i < n;
    const int64_t p = in ? (rows ? rows[i] : i) : 0;
    const uint32_t fl = in ? pods.flags[p] : 0u;
    const bool on = (fl & kPodValid) != 0;
    PodRegs<DT, LT, KEYS> r;
    load_pod<DT, LT, KEYS>(pods, p, r, true);
    if (!on) r.ns = 0;
    // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
    bool pod_err = on && !sp.ns_valid[r.ns];
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)r.ns * sp.gw;
    uint32_t cur_w = 0, cur_wi = 0xFFFFFFFFu;
    uint32_t n_exc = 0, n_act = 0, n_ins = 0;
    for (int t = 0; t < sp.T; ++t) {
      bool matched, err;
      walk_terms<LT, KEYS>(sp, t, ns_row, on, r.lp, r.lk, cur_w, cur_wi, matched, err);
      pod_err |= err;
      uint32_t st = 0;
      if (matched) {
        st = classify<DT>(recs + t, r.v, r.nzmask);
        n_exc += st == 4u;
        n_act += st == 2u;
        n_ins += st == 3u;
      }
      if (status && in) status[i * sp.T + t] = (uint8_t)st;
    }
    if (in) {
      summary[i] = on ? pack_summary(n_exc, n_act, n_ins, pod_err) : 0ull;
      if (status && pod_err)
        for (int t = 0; t < sp.T; ++t) status[i * sp.T + t] = 255;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The dense scans for label rows wider than 16 slots: same loop shape, the selector walk reads the pod's raw label
// row from HBM (walk_slow_mem) instead of holding it in registers.
// ---------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(kBlock) void kt_aggregate_dense_mem(PodTable pods, int64_t n_rows, SelProgram sp, unsigned long long* partial, int limb) {
  const int D = pods.D, stride = partial_stride(D);
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * kBlock) {
    const uint32_t fl = pods.flags[p];
    const bool countable = (fl & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
    if (!countable) continue;
    int64_t v[DT];
    load_requests<DT>(pods.req, pods.DS, p, v);
#pragma unroll
    for (int d = 0; d < DT; ++d) v[d] = limb_of(v[d], limb);
    const uint32_t present = fl >> kPresentShift;
    const bool not_finished = !(fl & kPodFinished);
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)pods.ns[p] * sp.gw;
    const uint32_t* lp = pods.lpair + p * pods.LS;
    const uint32_t* lk = pods.lkey + p * pods.LS;
    for (int t = 0; t < sp.T; ++t) {
      const uint32_t res = walk_slow_mem(sp, t, ns_row, true, lp, lk, pods.LS);
      unsigned long long* row = partial + (size_t)t * stride;
      if (res & kSlowError) atomicAdd(row + partial_off_errors(D), 1ull);
      if ((res & kSlowMatched) && not_finished) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
          if (d < D && ((present >> d) & 1u)) {
            if (v[d] != 0) atomicAdd(row + d, (unsigned long long)v[d]);
            atomicAdd(row + partial_off_presence(D) + d, 1ull);
          }
        atomicAdd(row + partial_off_pods(D), 1ull);
      }
    }
  }
}

template <int DT>
__global__ __launch_bounds__(kBlock) void kt_check_dense_mem(PodTable pods, int64_t n, const int64_t* rows, SelProgram sp,
                                                            const void* recs_, uint64_t* summary, uint8_t* status) {
  const CheckRec<DT>* recs = (const CheckRec<DT>*)recs_;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t p = rows ? rows[i] : i;
    const uint32_t fl = pods.flags[p];
    const bool on = (fl & kPodValid) != 0;
    const uint32_t ns = on ? pods.ns[p] : 0u;
    int64_t v[DT];
    load_requests<DT>(pods.req, pods.DS, p, v);
    uint32_t nzmask = 0;
#pragma unroll
    for (int d = 0; d < DT; ++d) nzmask |= (v[d] != 0 ? 1u : 0u) << d;
    bool pod_err = on && !sp.ns_valid[ns];
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)ns * sp.gw;
    const uint32_t* lp = pods.lpair + p * pods.LS;
    const uint32_t* lk = pods.lkey + p * pods.LS;
    uint32_t n_exc = 0, n_act = 0, n_ins = 0;
    for (int t = 0; t < sp.T; ++t) {
      const uint32_t res = walk_slow_mem(sp, t, ns_row, on, lp, lk, pods.LS);
      pod_err |= (res & kSlowError) != 0;
      uint32_t st = 0;
      if (res & kSlowMatched) {
        st = classify<DT>(recs + t, v, nzmask);
        n_exc += st == 4u;
        n_act += st == 2u;
        n_ins += st == 3u;
      }
      if (status) status[i * sp.T + t] = (uint8_t)st;
    }
    summary[i] = on ? pack_summary(n_exc, n_act, n_ins, pod_err) : 0ull;
    if (status && pod_err)
      for (int t = 0; t < sp.T; ++t) status[i * sp.T + t] = 255;
  }
}

#define KT_DISPATCH(KERNEL, DT_, LT_, KEYS_, ...)                                                              \
  do {                                                                                                         \
    if (DT_ == 4 && LT_ == 8 && !KEYS_) hipLaunchKernelGGL((KERNEL<4, 8, false>), __VA_ARGS__);                \
    else if (DT_ == 4 && LT_ == 8 && KEYS_) hipLaunchKernelGGL((KERNEL<4, 8, true>), __VA_ARGS__);             \
    else if (DT_ == 4 && LT_ == 16 && !KEYS_) hipLaunchKernelGGL((KERNEL<4, 16, false>), __VA_ARGS__);         \
    else if (DT_ == 4 && LT_ == 16 && KEYS_) hipLaunchKernelGGL((KERNEL<4, 16, true>), __VA_ARGS__);           \
    else if (DT_ == 8 && LT_ == 8 && !KEYS_) hipLaunchKernelGGL((KERNEL<8, 8, false>), __VA_ARGS__);           \
    else if (DT_ == 8 && LT_ == 8 && KEYS_) hipLaunchKernelGGL((KERNEL<8, 8, true>), __VA_ARGS__);             \
    else if (DT_ == 8 && LT_ == 16 && !KEYS_) hipLaunchKernelGGL((KERNEL<8, 16, false>), __VA_ARGS__);         \
    else if (DT_ == 8 && LT_ == 16 && KEYS_) hipLaunchKernelGGL((KERNEL<8, 16, true>), __VA_ARGS__);           \
    else if (DT_ == 16 && LT_ == 8 && !KEYS_) hipLaunchKernelGGL((KERNEL<16, 8, false>), __VA_ARGS__);         \
    else if (DT_ == 16 && LT_ == 8 && KEYS_) hipLaunchKernelGGL((KERNEL<16, 8, true>), __VA_ARGS__);           \
    else if (DT_ == 16 && LT_ == 16 && !KEYS_) hipLaunchKernelGGL((KERNEL<16, 16, false>), __VA_ARGS__);       \
    else hipLaunchKernelGGL((KERNEL<16, 16, true>), __VA_ARGS__);                                              \
  } while (0)

void launch_aggregate_dense(const PodTable& pods, int64_t n_rows, const SelProgram& sp, bool keys,
                            unsigned long long* partial, hipStream_t s, int limb) {
  if (n_rows <= 0 || sp.T <= 0) return;
  const int DT = dt_bucket(pods.D), LT = lt_bucket(pods.L);
  if (pods.LS > 16) {  // wide label rows: walked from HBM
    if (DT == 4) hipLaunchKernelGGL(kt_aggregate_dense_mem<4>, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, pods, n_rows, sp, partial, limb);
    else if (DT == 8) hipLaunchKernelGGL(kt_aggregate_dense_mem<8>, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, pods, n_rows, sp, partial, limb);
    else hipLaunchKernelGGL(kt_aggregate_dense_mem<16>, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, pods, n_rows, sp, partial, limb);
    return;
  }
  KT_DISPATCH(kt_aggregate_dense, DT, LT, keys, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, pods, n_rows, sp, partial, limb);
}

void launch_check_dense(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp, bool keys,
                        const void* recs, uint64_t* summary, uint8_t* status, hipStream_t s) {
  if (n <= 0) return;
  const int DT = dt_bucket(pods.D), LT = lt_bucket(pods.L);
  if (pods.LS > 16) {  // wide label rows: walked from HBM
    if (DT == 4) hipLaunchKernelGGL(kt_check_dense_mem<4>, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev, sp, recs, summary, status);
    else if (DT == 8) hipLaunchKernelGGL(kt_check_dense_mem<8>, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev, sp, recs, summary, status);
    else hipLaunchKernelGGL(kt_check_dense_mem<16>, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev, sp, recs, summary, status);
    return;
  }
  KT_DISPATCH(kt_check_dense, DT, LT, keys, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev, sp, recs, summary,
              status);
}

}  // namespace kt
