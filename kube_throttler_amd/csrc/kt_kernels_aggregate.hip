// kt_kernels_aggregate.hip — kt_aggregate_bitmap + kt_reduce_bitmap_slabs: per-throttle `used` through the bitmap index.
#include <cstdio>

#include "kt_bitmap_scan.h"

namespace kt {

__host__ __device__ inline uint32_t agg_bitmap_tab_bytes(int T, int D) { return (uint32_t)(((size_t)T * D * 8 + (size_t)T * 8 + 15) & ~(size_t)15); }

// compact argument block (see BmCheckArgs): the scalar register file only holds what the tile loop uses
constexpr uint32_t kAggListCap = 128;  // match-list entries per wave (512 B): the used-table needs the LDS
struct BmAggArgs {
  const uint32_t* ns;  // pod tables
  const uint32_t* flags;
  const int64_t* req;
  const uint32_t* lpair;
  const uint32_t* lkey;
  const SelProgram* sp;
  const uint32_t* slow_thr;
  unsigned long long* partial;
  unsigned char* slab;
  int64_t n_rows;
  BmIndexArgs ix;
  uint32_t off_list, off_pres, off_tab, tab_bytes;
  uint32_t n_slow;
  int32_t D, DS, LS, T;
};

static BmAggArgs make_bm_agg_args(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const SelProgram* sp_dev,
                                  const IndexDev& ix, unsigned long long* partial, unsigned char* slab, bool in_lds,
                                  uint32_t* total) {
  BmAggArgs a{};
  a.ns = pods.ns, a.flags = pods.flags, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.sp = sp_dev, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow, a.partial = partial, a.slab = slab, a.n_rows = n_rows;
  a.D = pods.D, a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_list = take((kBlockIx / kWave) * kAggListCap * 4);
  a.off_pres = take(kBlockIx * 2);
  a.tab_bytes = in_lds ? agg_bitmap_tab_bytes(sp.T, pods.D) : 0u;  // L2 form: no LDS table either (global atomics)
  a.off_tab = take(a.tab_bytes);
  plan_bitmap_index(ix, a.ix, in_lds, take);
  *total = o;
  return a;
}

// kt_aggregate_bitmap — `used` partials of this GPU's pod rows: affectedPods + fold Add for all throttles
// (throttle_controller.go:116-119,221-246; clusterthrottle_controller.go:119-122,224-270).
// INLDS: the bitmap index AND the workgroup's partial table live in LDS (small-T regime); otherwise the index is
// read through L2 and the amounts go straight to the partial buffer with global atomics.
// Wave-autonomous like kt_check_bitmap: lane = pod finds the tile's (pod, throttle) matches
// (bitmap_scan_tile), lane = (match, dimension pair) folds ResourceAmountOfPod into the workgroup's LDS
// table  tv i64[T][D] | tpres u32[T] (request-key presence mask) | tpods u32[T];  the table is spilled to
// this workgroup's slab at the end and kt_reduce_bitmap_slabs sums the slabs.
template <int DT, int LT, bool KEYS, bool INLDS>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_bitmap(const BmAggArgs a) {
  const int D = a.D, DS = a.DS, T = a.T;
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  lds_u64wp tv = (lds_u64wp)(lds + a.off_tab);
  lds_u32wp tpres = (lds_u32wp)(lds + a.off_tab + (uint32_t)T * D * 8);
  lds_u32wp tpods = tpres + T;
  for (uint32_t i = threadIdx.x; i < a.tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
  const BmView<INLDS> bm = open_bitmap_index<INLDS>(lds, a.ix);
  const int pstride = partial_stride(D);
  __syncthreads();
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  lds_u32wp list = (lds_u32wp)(lds + a.off_list) + wave * kAggListCap;
  KT_LDS uint16_t* l_pres = (KT_LDS uint16_t*)(lds + a.off_pres) + wave * kWave;  // [64] request-key presence masks of the tile
  constexpr int LPM = DT / 2, MPW = kWave / LPM;
  const uint32_t dp = lane % LPM, ml = lane / LPM;
  const bool dp_in = (int)(2 * dp) < DS;
  const uint32_t dpo = dp_in ? 2 * dp : 0u;
  const int64_t n_rows = a.n_rows;
  const int64_t n_wtiles = (n_rows + kWave - 1) / kWave;
  const int64_t wstep = (int64_t)gridDim.x * (kBlockIx / kWave);
  // A tile's selector-side records, always from valid addresses (lanes past the end re-read the last row and are
  // switched off), loaded ONE ROUND AHEAD; the request row is only touched (L2 prefetch for phase 2).
  struct Tile {
    uint32_t fl, ns;
    uint32_t lp[LT], lk[LT];
  };
  auto load_tile = [&](int64_t wt, Tile& t) {
    const int64_t p = min(wt * kWave + lane, n_rows - 1);
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
    const bool in = wt * kWave + lane < n_rows;
    const uint32_t fl = cur.fl;
    // shouldCountIn (throttle_controller.go:217-219); terminated pods are matched but not counted
    // (isNotFinished, pod_util.go:26-28) and only matter for error detection
    const bool countable = in && (fl & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
    const bool not_finished = !(fl & kPodFinished);
    const uint32_t ns = countable ? cur.ns : 0u;
    l_pres[lane] = (uint16_t)(fl >> kPresentShift);

    auto drain = [&](uint32_t n_items) {
      // ---- phase 2: lane = (match, dimension pair): fold the pod's amount into the table; operands are
      // fetched one step ahead of their use
      struct Ops {
        uint32_t vv, t, pres;
        kt_i64x2 x;
      };
      auto fetch = [&](uint32_t base, Ops& o) {
        const uint32_t j = base + ml;
        o.vv = j < n_items ? 1u : 0u;
        const uint32_t e = list[o.vv ? j : 0u];
        o.t = e & 0xFFFFFu;
        const uint32_t mp = (uint32_t)(wt * kWave) + (e >> 20);  // pod_capacity <= 2^31
        o.x = *(const kt_i64x2*)(a.req + (uint64_t)mp * (uint32_t)DS + dpo);
        o.pres = l_pres[e >> 20];
      };
      Ops c;
      fetch(0, c);
      for (uint32_t base = 0; base < n_items; base += MPW) {
        Ops nx;
        fetch(base + MPW, nx);
        if (c.vv && dp_in) {
          if (INLDS) {
            if (c.x.x != 0) lds_add64(tv + c.t * (uint32_t)D + 2 * dp, (unsigned long long)c.x.x);
            if (c.x.y != 0) lds_add64(tv + c.t * (uint32_t)D + 2 * dp + 1, (unsigned long long)c.x.y);  // padding dimension is 0
            if (dp == 0) {
              (void)__hip_atomic_fetch_or(tpres + c.t, c.pres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              lds_add(tpods + c.t, 1u);
            }
          } else {  // partial[t] = values[D] | presence counts[D] | pods | errors
            unsigned long long* pr = a.partial + (size_t)c.t * pstride;
            if (c.x.x != 0) atomicAdd(pr + 2 * dp, (unsigned long long)c.x.x);
            if (c.x.y != 0) atomicAdd(pr + 2 * dp + 1, (unsigned long long)c.x.y);
            // key presence: a positive value already makes the sum non-zero unless something negative cancels it,
            // and every non-positive contribution increments the presence word (kt_finalize: count != 0 || sum != 0)
            if (((c.pres >> (2 * dp)) & 1u) && c.x.x <= 0) atomicAdd(pr + D + 2 * dp, 1ull);
            if (((c.pres >> (2 * dp + 1)) & 1u) && c.x.y <= 0) atomicAdd(pr + D + 2 * dp + 1, 1ull);
            if (dp == 0) atomicAdd(pr + 2 * D, 1ull);
          }
        }
        c = nx;
      }
    };
    bitmap_scan_tile<LT, KEYS, kAggListCap>(bm, a.sp, a.slow_thr, a.n_slow, countable && not_finished, countable, ns, cur.lp,
                                            cur.lk, list, lane, drain, [&](uint32_t t) {  // rare: straight to the result buffer
                                              atomicAdd(a.partial + (size_t)t * pstride + 2 * D + 1, 1ull);
                                            });
    cur = nxt;
  }
  if (INLDS) {
    __syncthreads();  // spill this workgroup's table (coalesced 16-byte stores); kt_reduce_bitmap_slabs sums the slabs
    u32x4* dst = (u32x4*)(a.slab + (size_t)blockIdx.x * a.tab_bytes);
    lds_u4p src = (lds_u4p)(lds + a.off_tab);
    for (uint32_t i = threadIdx.x; i < a.tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
  }
}

// partial[t][j] = sum over slabs: j < D values; D <= j < 2D: key seen by the slab (0/1); j == 2D: pods.
// The error word (2D+1) is written by the scan kernel itself and left alone.
__global__ __launch_bounds__(1024) void kt_reduce_bitmap_slabs(const unsigned char* slab, int n_slabs, int T, int D,
                                                              unsigned long long* partial) {
  constexpr int G = 16;  // slab groups: every thread streams n_slabs / 16 independent loads
  __shared__ unsigned long long part[G][64];
  const int stride = partial_stride(D);
  const size_t pitch = agg_bitmap_tab_bytes(T, D);
  const int words = T * stride;
  const int wl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + wl;
  unsigned long long acc = 0;
  int j = 0;
  if (w < words) {
    const int t = w / stride;
    j = w - t * stride;
    if (j < D) {
      const unsigned char* base = slab + ((size_t)t * D + j) * 8;
#pragma unroll 16
      for (int b = g; b < n_slabs; b += G) acc += *(const unsigned long long*)(base + b * pitch);
    } else if (j < 2 * D) {
      const unsigned char* base = slab + (size_t)T * D * 8 + (size_t)t * 4;
#pragma unroll 16
      for (int b = g; b < n_slabs; b += G) acc += (*(const unsigned int*)(base + b * pitch) >> (j - D)) & 1u;
    } else if (j == 2 * D) {
      const unsigned char* base = slab + (size_t)T * D * 8 + (size_t)T * 4 + (size_t)t * 4;
#pragma unroll 16
      for (int b = g; b < n_slabs; b += G) acc += *(const unsigned int*)(base + b * pitch);
    }
  }
  part[g][wl] = acc;
  __syncthreads();
  if (g == 0 && w < words && j != 2 * D + 1) {
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) sum += part[k][wl];
    partial[w] = sum;
  }
}

#define KT_AGG_BM_CASE(DT_, LT_, KEYS_)                                                                        \
  {                                                                                                           \
    auto kfn = in_lds ? kt_aggregate_bitmap<DT_, LT_, KEYS_, true> : kt_aggregate_bitmap<DT_, LT_, KEYS_, false>; \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bm);     \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bm, s, bm_args);                                                      \
  }

static inline int agg_blocks(int64_t n_rows) {
  int64_t b = (n_rows + kBlockIx - 1) / kBlockIx;
  return (int)(b < 1 ? 1 : b > kCUs ? kCUs : b);
}

// slab scratch for the LDS-table form (0 when the table cannot live in LDS at all)
size_t aggregate_slab_bytes(int T, int D) {
  const size_t bytes = agg_bitmap_tab_bytes(T, D);
  if (bytes + 16 * 1024 > (size_t)kMaxLds) return 0;
  return (size_t)kCUs * bytes;
}

// `partial` must be zeroed by the caller.  Returns the dispatched scan kernel's symbol.
const char* launch_aggregate_indexed(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, bool keys, unsigned long long* partial, void* slab_, hipStream_t s,
                              const std::function<void()>& after_scan) {
  if (n_rows <= 0 || sp.T <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  unsigned char* slab = (unsigned char*)slab_;
  const int nb = agg_blocks(n_rows);
  dim3 g_(nb), b_(kBlockIx);
  uint32_t bm_total = 0;
  bool in_lds = slab != nullptr && aggregate_slab_bytes(sp.T, pods.D) != 0;
  BmAggArgs bm_args = make_bm_agg_args(pods, n_rows, sp, sp_dev, ix, partial, slab, in_lds, &bm_total);
  if (in_lds && bm_total > (uint32_t)kMaxLds) {
    in_lds = false;
    bm_args = make_bm_agg_args(pods, n_rows, sp, sp_dev, ix, partial, slab, false, &bm_total);
  }
  const size_t lds_bm = bm_total;
  static const bool dbg_lds = getenv("KT_DEBUG_LDS") != nullptr;
  if (dbg_lds) fprintf(stderr, "kt_aggregate_bitmap: in_lds=%d lds=%u blob=%u tab=%u T=%d\n", (int)in_lds, bm_total, ix.bm_blob_bytes, bm_args.tab_bytes, sp.T);
#ifdef KT_FAST_BUILD
  KT_AGG_BM_CASE(8, 8, false)
#else
  if (DT <= 8 && LT == 8) { if (keys) KT_AGG_BM_CASE(8, 8, true) else KT_AGG_BM_CASE(8, 8, false) }
  else if (DT <= 8) { if (keys) KT_AGG_BM_CASE(8, 16, true) else KT_AGG_BM_CASE(8, 16, false) }
  else if (LT == 8) { if (keys) KT_AGG_BM_CASE(16, 8, true) else KT_AGG_BM_CASE(16, 8, false) }
  else { if (keys) KT_AGG_BM_CASE(16, 16, true) else KT_AGG_BM_CASE(16, 16, false) }
#endif
  if (after_scan) after_scan();
  if (in_lds) {
    const int words = sp.T * partial_stride(pods.D);
    hipLaunchKernelGGL(kt_reduce_bitmap_slabs, dim3((words + 63) / 64), dim3(1024), 0, s, slab, nb, sp.T, pods.D, partial);
  }
  return in_lds ? "kt_aggregate_bitmap" : "kt_aggregate_bitmap_l2";
}

}  // namespace kt
