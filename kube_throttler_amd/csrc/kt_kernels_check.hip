// kt_kernels_check.hip — kt_check_bitmap: PreFilter for n pods through the bitmap form of the selector index (gfx950).
#include "kt_bitmap_scan.h"

namespace kt {

// ---------------------------------------------------------------------------------------------------
// kt_check_bitmap — PreFilter for n pods (plugin.go:148-215): CheckThrottled of both controllers through the
// bitmap form of the selector index (kt_index.h), LDS-resident when it fits (LDSIX; the small-T regime: up to a
// few thousand terms), else read through L2.
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
                                      uint64_t* summary, uint8_t* status, bool one_chunk, uint32_t* total) {
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
  a.off_rflags = one_chunk ? take((uint32_t)sp.T * 8) : 0u;  // several chunks: the flags are read through L2
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

// ONE: the whole program is one chunk (the small-T regime): throttle flags live in LDS and the summary is written
// once.  Otherwise the kernel walks the chunks: chunk image in, every tile of the workgroup scanned against it, the
// per-pod class counters carried from chunk to chunk in the summary words (finished by the last chunk).
template <int DT, int LT, bool KEYS, bool ONE>
__global__ __launch_bounds__(kBlockIx) void kt_check_bitmap(const BmCheckArgs a) {
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const CheckRec<DT>* recs = (const CheckRec<DT>*)a.recs;
  const u32x2* g_rflags = (const u32x2*)rec_flags<DT>((void*)a.recs, a.T);
  lds_stage16((KT_LDS u32x4*)(lds + a.ix.lds_buckets), a.ix.buckets, a.ix.bucket_bytes / 16u);
  if (ONE) {  // {flags, active_mask} of every throttle: 8 bytes each, rewritten by every kt_prepare_check
    KT_LDS u32x2* dst = (KT_LDS u32x2*)(lds + a.off_rflags);
    for (uint32_t i = threadIdx.x; i < (uint32_t)a.T; i += kBlockIx) dst[i] = g_rflags[i];
  }
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

  // A tile's selector-side records: 2 + LT/4 (+ LT/4) 128-bit loads per lane, issued back to back and always from
  // valid addresses (lanes past the end re-read the last pod and are switched off by `on`).  The request rows are
  // gathered by phase 2, for matched pods only.  (Loading tiles a round ahead, or touching the request rows early,
  // measurably does not help: the 16 waves of a CU already overlap each other's memory phases, and early touches
  // are evicted from L2 before they are used.)
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
  };
  const uint32_t n_chunks = ONE ? 1u : a.ix.n_chunks;
  for (uint32_t ci = 0; ci < n_chunks; ++ci) {
  const bool first = ci == 0, last = ci + 1 == n_chunks;
  __syncthreads();  // nobody reads the previous image any more
  const BmView bm = open_chunk(lds, a.ix, a.ix.chunks[ci]);
  __syncthreads();
  int64_t wt = (int64_t)blockIdx.x * (kBlockIx / kWave) + wave;
  for (; wt < n_wtiles; wt += wstep) {
    Tile cur;
    load_tile(wt, cur);
    // ---- phase 1: lane = pod
    const int64_t i = wt * kWave + lane;
    const bool in = i < n;
    // class counters so far (bit 1 = error) ride in the summary word between chunks
    const unsigned long long carried =
        (!ONE && !first && in) ? __hip_atomic_load((unsigned long long*)a.summary + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    cnt[lane] = carried & ~3ull;
    prow[lane] = cur.p;
    const bool on = in && (cur.fl & kPodValid) != 0;
    const uint32_t ns = on ? cur.ns : 0u;
    // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
    bool pod_err = (carried & 2ull) != 0 || (on & (a.ns_valid[ns] == 0));

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
        o.fa = ONE ? l_rflags[o.tt] : g_rflags[o.tt];  // {flags, active_mask}
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
    // throttles with unconvertible selectors are walked once, with the first chunk
    bitmap_scan_tile<LT, KEYS, kListCap, false>(bm, a.sp, a.slow_thr, first ? a.n_slow : 0u, on, on, ns, cur.lp, cur.lk,
                                                list, lane, drain, [&](uint32_t) { pod_err = true; }, [](uint32_t) {});
    // ---- phase 3: lane = pod
    if (in) {
      const unsigned long long c = cnt[lane];
      if (last) {
        a.summary[i] = !on ? 0ull : pod_err ? 2ull : (c | (c ? 1ull : 0ull));
        if (a.status && pod_err)
          for (int t = 0; t < a.T; ++t) a.status[i * a.T + t] = 255;
      } else {
        a.summary[i] = c | (pod_err ? 2ull : 0ull);
      }
    }
  }
  }
}

uint32_t check_fixed_lds() { return kBlockIx * 8 + (kBlockIx / kWave) * kListCap * 4 + kBlockIx * 4 + 64; }

#define KT_BM_CASE(DT_, LT_, KEYS_)                                                                            \
  {                                                                                                           \
    auto kfn = one ? kt_check_bitmap<DT_, LT_, KEYS_, true> : kt_check_bitmap<DT_, LT_, KEYS_, false>;         \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bytes, s, bm_args);                                                   \
  }

// returns the dispatched kernel's symbol, or nullptr when a chunk of the index does not fit the workgroup's LDS
// beside the working buffers (a single throttle with thousands of terms)
const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, bool keys, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s) {
  if (n <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  if (status) (void)hipMemsetAsync(status, 0, (size_t)n * (size_t)sp.T, s);
  int64_t nb = (n + kBlockIx - 1) / kBlockIx;
  if (nb > kCUs) nb = kCUs;
  dim3 g_((unsigned)nb), b_(kBlockIx);
  uint32_t bm_total = 0;
  bool one = ix.n_chunks == 1;
  BmCheckArgs bm_args = make_bm_check_args(pods, n, rows_dev, sp, sp_dev, ix, recs, summary, status, one, &bm_total);
  if (one && bm_total > (uint32_t)kMaxLds) {  // the flags of all throttles do not fit beside the image
    one = false;
    bm_args = make_bm_check_args(pods, n, rows_dev, sp, sp_dev, ix, recs, summary, status, false, &bm_total);
  }
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  const size_t lds_bytes = bm_total;
#ifdef KT_FAST_BUILD
  KT_BM_CASE(8, 8, false)
#else
  if (DT <= 8 && LT == 8) { if (keys) KT_BM_CASE(8, 8, true) else KT_BM_CASE(8, 8, false) }
  else if (DT <= 8) { if (keys) KT_BM_CASE(8, 16, true) else KT_BM_CASE(8, 16, false) }
  else if (LT == 8) { if (keys) KT_BM_CASE(16, 8, true) else KT_BM_CASE(16, 8, false) }
  else { if (keys) KT_BM_CASE(16, 16, true) else KT_BM_CASE(16, 16, false) }
#endif
  return one ? "kt_check_bitmap" : "kt_check_bitmap_chunked";
}

}  // namespace kt
