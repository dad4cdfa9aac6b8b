// kt_kernels_check.hip — kt_check_indexed: PreFilter for n pods through the label-atom index (gfx950).
#include "kt_bitmap_scan.h"

namespace kt {

// ---------------------------------------------------------------------------------------------------
// Wave-aggregated push of (pod_local, throttle) matches into a workgroup queue in LDS: one ds_add per
// wave per call instead of one per lane.  Returns false for a lane whose entry did not fit.
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool push_match(bool has, uint32_t entry, uint32_t* q, uint32_t* q_count) {
  const uint64_t mask = __ballot(has);
  if (mask == 0) return true;
  const uint32_t lane = __lane_id();
  const uint32_t leader = (uint32_t)__ffsll((unsigned long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(q_count, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (!has) return true;
  const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  if (pos < kQueueCap) {
    q[pos] = entry;
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------
// kt_check_indexed — PreFilter for n pods (plugin.go:148-215) through the index.
// Per 1024-pod tile: (1) lane = pod: probe the index, push matched (pod, throttle) pairs to the LDS queue;
// (2) lane = match: gather the pod's request vector (L2-hot) + the throttle's CheckRec, classify, bump the
// pod's class counters in LDS; (3) lane = pod: write the summary word.  Phase 2 runs with full lanes
// regardless of how unevenly matches are spread over pods.
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS, bool LDSIX>
__global__ __launch_bounds__(kBlockIx) void kt_check_indexed(PodTable pods, int64_t n, const int64_t* rows,
                                                            SelProgram sp, IndexDev ix, const void* recs_,
                                                            uint64_t* summary, uint8_t* status, int dbg) {
  const CheckRec<DT>* recs = (const CheckRec<DT>*)recs_;
  // LDS carve: [cnt u64 x 1024][queue u32 x kQueueCap][q_count][index copy ...]
  unsigned long long* cnt = (unsigned long long*)kt_smem;
  uint32_t* q = (uint32_t*)(kt_smem + kBlockIx * 8);
  uint32_t* q_count = q + kQueueCap;
  unsigned char* ix_base = kt_smem + kBlockIx * 8 + kQueueCap * 4 + 16;
  if (LDSIX) {  // stage hash slots + postings into LDS once per workgroup (16-byte copies)
    const uint4* src_s = (const uint4*)ix.slots;
    uint4* dst = (uint4*)ix_base;
    const uint32_t ns16 = ix.n_slots, np16 = ix.n_postings * 2;
    for (uint32_t i = threadIdx.x; i < ns16; i += kBlockIx) dst[i] = src_s[i];
    const uint4* src_p = (const uint4*)ix.postings;
    for (uint32_t i = threadIdx.x; i < np16; i += kBlockIx) dst[ns16 + i] = src_p[i];
  }
  lds_u4p l_slots = (lds_u4p)(KT_LDS unsigned char*)ix_base;
  lds_u4p l_posts = l_slots + ix.n_slots;
  const int64_t n_tiles = (n + kBlockIx - 1) / kBlockIx;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t i = tile * kBlockIx + threadIdx.x;
    cnt[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) *q_count = 0u;
    __syncthreads();
    // ---- phase 1: lane = pod
    const bool in = i < n;
    const int64_t p = in ? (rows ? rows[i] : i) : 0;
    const uint32_t fl = in ? pods.flags[p] : 0u;
    const bool on = (fl & kPodValid) != 0;
    bool pod_err = false;
    if (on) {
      uint32_t lp[LT], lk[LT];
      const uint32_t ns = pods.ns[p];
      load_labels<LT, KEYS>(pods.lpair, pods.lkey, pods.LS, p, lp, lk);
      // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
      pod_err = !sp.ns_valid[ns];
      auto on_match = [&](uint32_t t) {
        if (push_match(true, (uint32_t)threadIdx.x << 20 | t, q, q_count)) return;
        // queue full: classify inline (rare)
        int64_t v[DT];
        uint32_t nz = 0;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          v[d] = d < pods.D ? pods.req[(int64_t)p * pods.DS + d] : 0;
          nz |= (v[d] != 0 ? 1u : 0u) << d;
        }
        const uint32_t st = classify<DT>(recs + t, v, nz);
        if (st != 1u) atomicAdd(cnt + threadIdx.x, st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
        if (status) status[i * sp.T + t] = (uint8_t)st;
      };
      if (dbg != 2) {
      if (LDSIX) enumerate_matches<LT, KEYS>(sp, ix, l_slots, l_posts, ns, lp, lk, on_match);
      else enumerate_matches<LT, KEYS>(sp, ix, (const u32x4*)ix.slots, (const u32x4*)ix.postings, ns, lp, lk, on_match);
      }
      const uint32_t* ns_row = sp.ns_term_ok + (size_t)ns * sp.gw;
      for (uint32_t k = 0; k < ix.n_slow; ++k) {
        bool matched, err;
        const int t = (int)ix.slow_thr[k];
        walk_slow<LT, KEYS>(sp, t, ns_row, true, lp, lk, matched, err);
        pod_err |= err;
        if (matched) on_match((uint32_t)t);
      }
      if (dbg == 2) pod_err |= (lp[0] ^ lp[LT - 1] ^ lk[0]) == 0xFFFFFFFFu;
    }
    __syncthreads();
    // ---- phase 2: lane = (match, dimension): DT lanes share one match, so the pod's request row and the
    // throttle's thr[] / head[] rows are each ONE coalesced transaction per match
    const uint32_t qn = dbg == 1 ? 0u : min(*q_count, kQueueCap);
    {
      constexpr int MPW = kWave / DT;  // matches per wave per iteration
      const uint32_t lane = threadIdx.x & (kWave - 1), d = lane % DT, ml = lane / DT;
      const uint32_t wave = threadIdx.x / kWave;
      const uint64_t gmask = (DT == 64 ? ~0ull : ((1ull << DT) - 1ull)) << (ml * DT);
      for (uint32_t base = wave * MPW; base < qn; base += (kBlockIx / kWave) * MPW) {
        const uint32_t j = base + ml;
        const bool valid = j < qn;
        const uint32_t e = valid ? q[j] : 0u;
        const uint32_t pl = e >> 20, t = e & 0xFFFFFu;
        const int64_t mi = tile * kBlockIx + pl;
        const int64_t mp = rows ? rows[valid ? mi : 0] : mi;
        const CheckRec<DT>* rc = recs + t;
        const int64_t v = (valid && (int)d < pods.D) ? pods.req[(int64_t)mp * pods.DS + d] : 0;
        const bool nz = v != 0;
        const uint32_t amask = rc->active_mask;
        const bool exc_d = valid && nz && v > rc->thr[d];
        const bool ins_d = valid && nz && v > rc->head[d];
        const bool act_d = valid && nz && ((amask >> d) & 1u);
        const uint64_t be = __ballot(exc_d), ba = __ballot(act_d), bi = __ballot(ins_d);
        if (valid && d == 0) {
          const uint32_t f = rc->flags;
          const bool exc = (f & kRecExceedsByCount) || (be & gmask);
          const bool act = (f & kRecActiveByCount) || (ba & gmask);
          const bool ins = (f & kRecInsufficientByCount) || (bi & gmask);
          const uint32_t st = exc ? 4u : act ? 2u : ins ? 3u : 1u;
          if (st != 1u) atomicAdd(cnt + pl, st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
          if (status) status[mi * sp.T + t] = (uint8_t)st;
        }
      }
    }
    __syncthreads();
    // ---- phase 3: lane = pod
    if (in) {
      const unsigned long long c = cnt[threadIdx.x];
      summary[i] = !on ? 0ull : pod_err ? 2ull : (c | (c ? 1ull : 0ull));
      if (status && pod_err)
        for (int t = 0; t < sp.T; ++t) status[i * sp.T + t] = 255;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_check_bitmap — same contract as kt_check_indexed, for selector programs whose bitmap form
// (kt_index.h) fits in LDS (the small-T regime: up to a few thousand terms).
// WAVE-AUTONOMOUS: after the one-time staging of the tables, every wave walks its own 64-pod tiles and
// never meets a workgroup barrier again — all of a tile's state (class counters, match list, request
// rows) belongs to the wave that owns the 64 pods.
//   phase 1, lane = pod: the pod's record arrives as 128-bit row loads (2 for 8 labels, 4 for 8 request
//            dimensions) issued back to back; bitmap_scan_tile (kt_bitmap_scan.h) turns it into the
//            tile's dense match list.
//   phase 2, lane = (match, dimension pair): CheckThrottledFor through the CheckRec algebra; the pod's
//            request row comes from the wave's LDS tile, thr[] / head[] as 16-byte pieces from L2.
//   phase 3, lane = pod: the 8-byte summary word.
// ---------------------------------------------------------------------------------------------------

// Everything the kernel needs, and nothing else: a compact argument block keeps the scalar register file
// free of the (large) generic table descriptors, which are reached through `sp` only on rare paths.
constexpr uint32_t kListCap = 256;  // match-list entries per wave (1 KB)
struct BmCheckArgs {
  const uint32_t* ns;  // pod tables
  const uint32_t* flags;
  const int64_t* req;
  const uint32_t* lpair;
  const uint32_t* lkey;
  const int64_t* rows;
  const void* recs;
  uint64_t* summary;
  uint8_t* status;
  const SelProgram* sp;  // device copy (rare term shapes, slow throttles)
  const uint8_t* ns_valid;
  const uint32_t* slow_thr;
  int64_t n;
  BmIndexArgs ix;
  uint32_t off_cnt, off_list, off_req, off_rflags;
  uint32_t n_slow;
  int32_t DS, LS, T;
};

static BmCheckArgs make_bm_check_args(const PodTable& pods, int64_t n, const int64_t* rows, const SelProgram& sp,
                                      const SelProgram* sp_dev, const IndexDev& ix, const void* recs,
                                      uint64_t* summary, uint8_t* status, uint32_t* total) {
  BmCheckArgs a{};
  a.ns = pods.ns, a.flags = pods.flags, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.n = n, a.rows = rows, a.recs = recs, a.summary = summary, a.status = status;
  a.sp = sp_dev, a.ns_valid = sp.ns_valid, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow;
  a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_cnt = take(kBlockIx * 8);
  a.off_list = take((kBlockIx / kWave) * kListCap * 4);
  a.off_req = take(kBlockIx * 4);
  a.off_rflags = take((uint32_t)sp.T * 8);
  plan_bitmap_index(ix, a.ix, take);
  *total = o;
  return a;
}

// OR over the LPM consecutive lanes of a match group (LPM = 4 or 8): data-parallel primitives, no LDS traffic
template <int LPM>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  if (LPM == 8) v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  return v;
}

template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_check_bitmap(const BmCheckArgs a) {
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const BmView bm = stage_bitmap_index(lds, a.ix);
  const CheckRec<DT>* recs = (const CheckRec<DT>*)a.recs;
  {  // {flags, active_mask} of every throttle: 8 bytes each, rewritten by every kt_prepare_check
    const u32x2* src = (const u32x2*)rec_flags<DT>((void*)a.recs, a.T);
    KT_LDS u32x2* dst = (KT_LDS u32x2*)(lds + a.off_rflags);
    for (uint32_t i = threadIdx.x; i < (uint32_t)a.T; i += kBlockIx) dst[i] = src[i];
  }
  __syncthreads();  // the only workgroup barrier
  const KT_LDS u32x2* l_rflags = (const KT_LDS u32x2*)(lds + a.off_rflags);
  const int64_t n = a.n;
  const int DS = a.DS;
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  // this wave's private LDS areas
  lds_u64wp cnt = (lds_u64wp)(lds + a.off_cnt) + wave * kWave;         // [64] class counters
  lds_u32wp list = (lds_u32wp)(lds + a.off_list) + wave * kListCap;    // match list
  lds_u32wp prow = (lds_u32wp)(lds + a.off_req) + wave * kWave;       // [64] pod table rows of the tile
  // phase-2 lane mapping: LPM lanes per match, two dimensions each
  constexpr int LPM = DT / 2, MPW = kWave / LPM;
  const uint32_t dp = lane % LPM, ml = lane / LPM;
  const bool dp_in = (int)(2 * dp) < DS;
  const uint32_t dpo = dp_in ? 2 * dp : 0u;
  const int64_t n_wtiles = (n + kWave - 1) / kWave;
  const int64_t wstep = (int64_t)gridDim.x * (kBlockIx / kWave);

  // A tile's selector-side records: 2 + LT/4 (+ LT/4) loads per lane, always from valid addresses (lanes past
  // the end re-read the last pod and are switched off by `on`), loaded ONE ROUND AHEAD so that a round never
  // starts by waiting for HBM.  The request row of the next round is only touched (L2 prefetch): it is loaded
  // at the start of its own round and parked in the wave's LDS tile for phase 2.
  struct Tile {
    uint32_t fl, ns, p;
    uint32_t lp[LT], lk[LT];
  };
  auto load_tile = [&](int64_t wt, Tile& t) {
    const int64_t i = min(wt * kWave + lane, n - 1);
    const int64_t p = a.rows ? a.rows[i] : i;
    t.p = (uint32_t)p;  // pod_capacity <= 2^31
    t.fl = a.flags[p];
    t.ns = a.ns[p];
    load_labels<LT, KEYS>(a.lpair, a.lkey, a.LS, p, t.lp, t.lk);
    (void)*(const volatile uint32_t*)(a.req + p * DS);
  };
  int64_t wt = (int64_t)blockIdx.x * (kBlockIx / kWave) + wave;
  Tile cur;
  if (wt < n_wtiles) load_tile(wt, cur);
  for (; wt < n_wtiles; wt += wstep) {
    Tile nxt;
    load_tile(min(wt + wstep, n_wtiles - 1), nxt);
    // ---- phase 1: lane = pod
    const int64_t i = wt * kWave + lane;
    const bool in = i < n;
    cnt[lane] = 0ull;
    prow[lane] = cur.p;
    const bool on = in && (cur.fl & kPodValid) != 0;
    const uint32_t ns = on ? cur.ns : 0u;
    // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
    bool pod_err = on & (a.ns_valid[ns] == 0);

    auto drain = [&](uint32_t n_items) {
      // ---- phase 2: lane = (match, dimension pair); operands are fetched one step ahead of their use
      struct Ops {
        uint32_t vv, pl, tt;
        u32x2 fa;
        kt_i64x2 xx, th, hd;
      };
      auto fetch = [&](uint32_t base, Ops& o) {
        const uint32_t j = base + ml;
        o.vv = j < n_items ? 1u : 0u;
        const uint32_t e = list[o.vv ? j : 0u];
        o.pl = e >> 20;
        o.tt = e & 0xFFFFFu;
        const CheckRec<DT>* rc = recs + o.tt;
        o.th = *(const kt_i64x2*)(rc->thr + 2 * dp);
        o.hd = *(const kt_i64x2*)(rc->head + 2 * dp);
        o.fa = l_rflags[o.tt];  // {flags, active_mask}
        o.xx = *(const kt_i64x2*)(a.req + (uint64_t)prow[o.pl] * (uint32_t)DS + dpo);
      };
      Ops c;
      fetch(0, c);
      for (uint32_t base = 0; base < n_items; base += MPW) {
        Ops nx;
        fetch(base + MPW, nx);
        const bool live = c.vv && dp_in;
        const bool nz0 = live && c.xx.x != 0, nz1 = live && c.xx.y != 0;
        const uint32_t am = c.fa.y >> (2 * dp);
        uint32_t bits = ((nz0 && c.xx.x > c.th.x) || (nz1 && c.xx.y > c.th.y)) ? 1u : 0u;  // exceeds
        bits |= ((nz0 && (am & 1u)) || (nz1 && (am & 2u))) ? 2u : 0u;                       // active
        bits |= ((nz0 && c.xx.x > c.hd.x) || (nz1 && c.xx.y > c.hd.y)) ? 4u : 0u;          // insufficient
        bits = group_or<LPM>(bits);
        const uint32_t ff = c.fa.x;
        const bool exc = (ff & kRecExceedsByCount) || (bits & 1u);
        const bool act = (ff & kRecActiveByCount) || (bits & 2u);
        const bool ins = (ff & kRecInsufficientByCount) || (bits & 4u);
        const uint32_t st = exc ? 4u : act ? 2u : ins ? 3u : 1u;
        if (dp == 0 && c.vv) {
          if (st != 1u) lds_add64(cnt + c.pl, st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
          if (a.status) a.status[(wt * kWave + c.pl) * a.T + c.tt] = (uint8_t)st;
        }
        c = nx;
      }
    };
    bitmap_scan_tile<LT, KEYS, kListCap>(bm, a.sp, a.slow_thr, a.n_slow, on, on, ns, cur.lp, cur.lk, list, lane, drain,
                               [&](uint32_t) { pod_err = true; });
    // ---- phase 3: lane = pod
    if (in) {
      const unsigned long long c = cnt[lane];
      a.summary[i] = !on ? 0ull : pod_err ? 2ull : (c | (c ? 1ull : 0ull));
      if (a.status && pod_err)
        for (int t = 0; t < a.T; ++t) a.status[i * a.T + t] = 255;
    }
    cur = nxt;
  }
}

#define KT_BM_CASE(DT_, LT_, KEYS_)                                                                            \
  {                                                                                                           \
    auto kfn = kt_check_bitmap<DT_, LT_, KEYS_>;                                                              \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bytes, s, bm_args);                                                   \
  }

const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, bool keys, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s) {
  if (n <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  if (status) (void)hipMemsetAsync(status, 0, (size_t)n * (size_t)sp.T, s);
  int64_t nb = (n + kBlockIx - 1) / kBlockIx;
  if (nb > kCUs) nb = kCUs;
  dim3 g_((unsigned)nb), b_(kBlockIx);
  static const int dbg = getenv("KT_DEBUG_MODE") ? atoi(getenv("KT_DEBUG_MODE")) : 0;
  // small-T regime: the whole selector program as LDS-resident bitmaps
  if (ix.bm_words != 0 && dbg != 3) {
    uint32_t bm_total = 0;
    const BmCheckArgs bm_args = make_bm_check_args(pods, n, rows_dev, sp, sp_dev, ix, recs, summary, status, &bm_total);
    if (bm_total <= (uint32_t)kMaxLds) {
      const size_t lds_bytes = bm_total;
#ifdef KT_FAST_BUILD
      KT_BM_CASE(8, 8, false)
#else
      if (DT <= 8 && LT == 8) { if (keys) KT_BM_CASE(8, 8, true) else KT_BM_CASE(8, 8, false) }
      else if (DT <= 8) { if (keys) KT_BM_CASE(8, 16, true) else KT_BM_CASE(8, 16, false) }
      else if (LT == 8) { if (keys) KT_BM_CASE(16, 8, true) else KT_BM_CASE(16, 8, false) }
      else { if (keys) KT_BM_CASE(16, 16, true) else KT_BM_CASE(16, 16, false) }
#endif
      return "kt_check_bitmap";
    }
  }
  const size_t ix_bytes = (size_t)ix.n_slots * sizeof(IndexSlot) + (size_t)ix.n_postings * sizeof(Posting);
  const size_t fixed_bytes = kBlockIx * 8 + kQueueCap * 4 + 16;
  const bool lds_ix = ix_bytes + fixed_bytes <= (size_t)kMaxLds && n >= 4 * kBlockIx;
  const size_t lds_bytes = fixed_bytes + (lds_ix ? ((ix_bytes + 15) & ~(size_t)15) : 0);
#define KT_IX_ARGS pods, n, rows_dev, sp, ix, recs, summary, status, dbg
  if (lds_ix) KT_IX_DISPATCH2(kt_check_indexed, DT, LT, keys, true);
  else KT_IX_DISPATCH2(kt_check_indexed, DT, LT, keys, false);
#undef KT_IX_ARGS
  return "kt_check_indexed";
}

}  // namespace kt
