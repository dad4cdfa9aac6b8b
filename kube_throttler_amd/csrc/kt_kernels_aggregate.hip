// kt_kernels_aggregate.hip — kt_aggregate_bitmap + kt_reduce_bitmap_slabs: per-throttle `used` through the bitmap index.
#include <cstdio>

#include "kt_bitmap_scan.h"

namespace kt {

// LDS table of one chunk, 16-byte granules:
//   tv i64[n_thr][D] | tpres u32[n_thr] (request-key presence MASK) | tpods u32[n_thr]
//   counts mode (incremental engines): tv i64[n_thr][D] | tcnt u32[n_thr][D] (pods carrying the key) | tpods u32[n_thr]
__host__ __device__ inline uint32_t agg_tab_bytes(uint32_t n_thr, int D, bool counts) {
  const size_t per = (size_t)D * 8 + (counts ? (size_t)D * 4 + 4 : 8);
  return (uint32_t)(((size_t)n_thr * per + 15) & ~(size_t)15);
}

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
  unsigned long long* partial;  // target of the slow-list atomics (and of the slab reduction)
  unsigned char* slab;
  const int64_t* rows;  // nullable: the pods to scan are rows[0..n_rows) instead of row0 + [0, n_rows)
  int64_t row0, n_rows;
  BmIndexArgs ix;
  uint32_t off_list, off_pres, off_tab;
  uint32_t n_slow;
  int32_t D, DS, LS, T;
  int32_t counts;  // table keeps per-key pod counts instead of the presence mask
  int32_t sign;    // +1 / -1: the scanned pods are added to / removed from the target (delta scans)
};

static BmAggArgs make_bm_agg_args(const PodTable& pods, const AggScan& sc, const SelProgram& sp, const SelProgram* sp_dev,
                                  const IndexDev& ix, unsigned long long* partial, unsigned char* slab, uint32_t* total) {
  const int64_t n_rows = sc.n;
  BmAggArgs a{};
  a.rows = sc.rows, a.row0 = sc.row0, a.counts = sc.counts ? 1 : 0, a.sign = sc.sign;
  a.ns = pods.ns, a.flags = pods.flags, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.sp = sp_dev, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow, a.partial = partial, a.slab = slab, a.n_rows = n_rows;
  a.D = pods.D, a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_list = take((kBlockIx / kWave) * kAggListCap * 4);
  a.off_pres = take(kBlockIx * 2 + kBlockIx * 4);  // presence masks u16[1024] | pod rows u32[1024]
  a.off_tab = take(agg_tab_bytes(ix.bm_max_thr, pods.D, sc.counts));
  plan_bitmap_index(ix, a.ix, take);
  *total = o;
  return a;
}

// kt_aggregate_bitmap — `used` partials of this GPU's pod rows: affectedPods + fold Add for all throttles
// (throttle_controller.go:116-119,221-246; clusterthrottle_controller.go:119-122,224-270).
// The workgroup walks the chunks of the index: chunk image in, table of the chunk's throttles zeroed, every tile of
// the workgroup scanned against it — wave-autonomous like kt_check_bitmap: lane = pod finds the tile's
// (pod, throttle) matches (bitmap_scan_tile), lane = (match, dimension pair) folds ResourceAmountOfPod into the
// LDS table  tv i64[n_thr][D] | tpres u32[n_thr] (request-key presence mask) | tpods u32[n_thr] — then the table is
// spilled to this (chunk, workgroup)'s slab; kt_reduce_bitmap_slabs sums the slabs into the partial buffer.
// No global atomics except for throttles with unconvertible selectors (the "slow" list).
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_bitmap(const BmAggArgs a) {
  const int D = a.D, DS = a.DS;
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const int pstride = partial_stride(D);
  lds_stage16((KT_LDS u32x4*)(lds + a.ix.lds_buckets), a.ix.buckets, a.ix.bucket_bytes / 16u);
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
  // switched off).  The request rows are gathered by phase 2 for matched pods only.
  struct Tile {
    uint32_t fl, ns, p;
    uint32_t lp[LT], lk[LT];
  };
  auto load_tile = [&](int64_t wt, Tile& t) {
    const int64_t i = min(wt * kWave + lane, n_rows - 1);
    const int64_t p = a.rows ? a.rows[i] : a.row0 + i;
    t.p = (uint32_t)p;  // pod_capacity <= 2^31
    t.fl = a.flags[p];
    t.ns = a.ns[p];
    load_labels<LT, KEYS>(a.lpair, a.lkey, a.LS, p, t.lp, t.lk);
  };
  for (uint32_t ci = 0; ci < a.ix.n_chunks; ++ci) {
    const BmChunk ch = a.ix.chunks[ci];
    const uint32_t n_thr = ch.n_thr;
    const bool counts = a.counts != 0;
    const uint32_t tab_bytes = agg_tab_bytes(n_thr, D, counts);
    lds_u64wp tv = (lds_u64wp)(lds + a.off_tab);
    lds_u32wp tpres = (lds_u32wp)(lds + a.off_tab + n_thr * (uint32_t)D * 8);  // mask [n] or counts [n][D]
    lds_u32wp tpods = tpres + (counts ? n_thr * (uint32_t)D : n_thr);
    __syncthreads();  // nobody reads the previous image / table any more
    for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
    const BmView bm = open_chunk(lds, a.ix, ch);
    __syncthreads();
    int64_t wt = (int64_t)blockIdx.x * (kBlockIx / kWave) + wave;
    for (; wt < n_wtiles; wt += wstep) {
      Tile cur;
      load_tile(wt, cur);
      // ---- phase 1: lane = pod
      const bool in = wt * kWave + lane < n_rows;
      const uint32_t fl = cur.fl;
      // shouldCountIn (throttle_controller.go:217-219); terminated pods are matched but not counted
      // (isNotFinished, pod_util.go:26-28) and only matter for error detection
      const bool countable = in && (fl & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
      const bool not_finished = !(fl & kPodFinished);
      const uint32_t ns = countable ? cur.ns : 0u;
      const uint32_t present = fl >> kPresentShift;
      l_pres[lane] = (uint16_t)present;
      const uint32_t mp_lane = cur.p;
      KT_LDS uint32_t* l_row = (KT_LDS uint32_t*)(lds + a.off_pres + kBlockIx * 2) + wave * kWave;  // [64] pod rows of the tile
      l_row[lane] = mp_lane;

      auto drain = [&](uint32_t n_items) {
        // ---- phase 2: lane = (match, dimension pair): fold the pod's amount into the table; operands are
        // fetched one step ahead of their use
        struct Ops {
          uint32_t vv, r, pres;
          kt_i64x2 x;
        };
        auto fetch = [&](uint32_t base, Ops& o) {
          const uint32_t j = base + ml;
          o.vv = j < n_items ? 1u : 0u;
          const uint32_t e = list[o.vv ? j : 0u];
          o.r = e & 0xFFFFFu;  // chunk-local throttle rank
          const uint32_t mp = l_row[e >> 20];
          o.x = *(const kt_i64x2*)(a.req + (uint64_t)mp * (uint32_t)DS + dpo);
          o.pres = l_pres[e >> 20];
        };
        Ops c;
        fetch(0, c);
        for (uint32_t base = 0; base < n_items; base += MPW) {
          Ops nx;
          fetch(base + MPW, nx);
          if (c.vv && dp_in) {
            if (c.x.x != 0) lds_add64(tv + c.r * (uint32_t)D + 2 * dp, (unsigned long long)c.x.x);
            if (c.x.y != 0) lds_add64(tv + c.r * (uint32_t)D + 2 * dp + 1, (unsigned long long)c.x.y);  // padding dimension is 0
            if (counts) {
              if ((c.pres >> (2 * dp)) & 1u) lds_add(tpres + c.r * (uint32_t)D + 2 * dp, 1u);
              if ((c.pres >> (2 * dp + 1)) & 1u) lds_add(tpres + c.r * (uint32_t)D + 2 * dp + 1, 1u);
            } else if (dp == 0) {
              (void)__hip_atomic_fetch_or(tpres + c.r, c.pres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (dp == 0) lds_add(tpods + c.r, 1u);
          }
          c = nx;
        }
      };
      // throttles with unconvertible selectors have no rank: walked once (with the first chunk), straight to the
      // result buffer
      bitmap_scan_tile<LT, KEYS, kAggListCap, true>(
          bm, a.sp, a.slow_thr, ci == 0 ? a.n_slow : 0u, countable && not_finished, countable, ns, cur.lp, cur.lk, list,
          lane, drain,
          [&](uint32_t t) { atomicAdd(a.partial + (size_t)t * pstride + 2 * D + 1, (unsigned long long)(long long)a.sign); },
          [&](uint32_t t) {
            unsigned long long* pr = a.partial + (size_t)t * pstride;
            for (int d = 0; d < D; ++d)
              if ((present >> d) & 1u) {
                const int64_t v = a.req[(uint64_t)mp_lane * (uint32_t)DS + d];
                if (v != 0) atomicAdd(pr + d, (unsigned long long)(a.sign * v));
                atomicAdd(pr + D + d, (unsigned long long)(long long)a.sign);
              }
            atomicAdd(pr + 2 * D, (unsigned long long)(long long)a.sign);
          });
    }
    __syncthreads();  // spill this (chunk, workgroup)'s table: coalesced 16-byte stores
    u32x4* dst = (u32x4*)(a.slab + (size_t)ch.slab_off * 16 + (size_t)blockIdx.x * tab_bytes);
    lds_u4p src = (lds_u4p)(lds + a.off_tab);
    for (uint32_t i = threadIdx.x; i < tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
  }
}

// partial[t][j] += sum over the workgroups' slabs of t's chunk: j < D values; D <= j < 2D: key seen by the slab
// (0/1); j == 2D: pods.  "+=": throttles of the slow list were written by the scan kernel with atomics, and so was
// every error word (2D+1), which is left alone.  grid = (word groups, chunks).
__global__ __launch_bounds__(1024) void kt_reduce_bitmap_slabs(const unsigned char* slab, const BmChunk* chunks,
                                                              const uint32_t* rank_t, int n_slabs, int D, int counts, int sign,
                                                              unsigned long long* partial) {
  constexpr int G = 16;  // slab groups: every thread streams n_slabs / 16 independent loads
  __shared__ unsigned long long part[G][64];
  const BmChunk ch = chunks[blockIdx.y];
  const int n_thr = (int)ch.n_thr;
  const int stride = partial_stride(D);
  const size_t pitch = agg_tab_bytes(ch.n_thr, D, counts != 0);
  const unsigned char* base0 = slab + (size_t)ch.slab_off * 16;
  const int words = n_thr * stride;
  if ((int)blockIdx.x * 64 >= words) return;
  const int wl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + wl;
  unsigned long long acc = 0;
  int j = 0, r = 0;
  if (w < words) {
    r = w / stride;
    j = w - r * stride;
    if (j < D) {
      const unsigned char* base = base0 + ((size_t)r * D + j) * 8;
#pragma unroll 16
      for (int b = g; b < n_slabs; b += G) acc += *(const unsigned long long*)(base + b * pitch);
    } else if (j < 2 * D) {
      if (counts) {
        const unsigned char* base = base0 + (size_t)n_thr * D * 8 + ((size_t)r * D + (j - D)) * 4;
#pragma unroll 16
        for (int b = g; b < n_slabs; b += G) acc += *(const unsigned int*)(base + b * pitch);
      } else {
        const unsigned char* base = base0 + (size_t)n_thr * D * 8 + (size_t)r * 4;
#pragma unroll 16
        for (int b = g; b < n_slabs; b += G) acc += (*(const unsigned int*)(base + b * pitch) >> (j - D)) & 1u;
      }
    } else if (j == 2 * D) {
      const unsigned char* base = base0 + (size_t)n_thr * D * 8 + (size_t)n_thr * 4 * (counts ? D : 1) + (size_t)r * 4;
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
    partial[(size_t)rank_t[ch.rank0 + r] * stride + j] += (unsigned long long)((long long)sign * (long long)sum);
  }
}

#define KT_AGG_BM_CASE(DT_, LT_, KEYS_)                                                                        \
  {                                                                                                           \
    auto kfn = kt_aggregate_bitmap<DT_, LT_, KEYS_>;                                                          \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bm);     \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bm, s, bm_args);                                                      \
  }

static inline int agg_blocks(int64_t n_rows) {
  int64_t b = (n_rows + kBlockIx - 1) / kBlockIx;
  return (int)(b < 1 ? 1 : b > kCUs ? kCUs : b);
}

uint32_t aggregate_fixed_lds() { return (kBlockIx / kWave) * kAggListCap * 4 + kBlockIx * 2 + kBlockIx * 4 + 64; }

// `partial` must be zeroed by the caller.  Returns the dispatched scan kernel's symbol, nullptr when a chunk of the
// index does not fit the workgroup's LDS.
const char* launch_aggregate_indexed(const PodTable& pods, const AggScan& sc, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, bool keys, unsigned long long* partial, void* slab_, hipStream_t s,
                              const std::function<void()>& after_scan) {
  const int64_t n_rows = sc.n;
  if (n_rows <= 0 || sp.T <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  unsigned char* slab = (unsigned char*)slab_;
  const int nb = agg_blocks(n_rows);
  dim3 g_(nb), b_(kBlockIx);
  uint32_t bm_total = 0;
  const BmAggArgs bm_args = make_bm_agg_args(pods, sc, sp, sp_dev, ix, partial, slab, &bm_total);
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  const size_t lds_bm = bm_total;
  static const bool dbg_lds = getenv("KT_DEBUG_LDS") != nullptr;
  if (dbg_lds) fprintf(stderr, "kt_aggregate_bitmap: lds=%u chunks=%u max image=%u max thr=%u T=%d\n", bm_total, ix.n_chunks, ix.bm_max_img, ix.bm_max_thr, sp.T);
#ifdef KT_FAST_BUILD
  KT_AGG_BM_CASE(8, 8, false)
#else
  if (DT <= 8 && LT == 8) { if (keys) KT_AGG_BM_CASE(8, 8, true) else KT_AGG_BM_CASE(8, 8, false) }
  else if (DT <= 8) { if (keys) KT_AGG_BM_CASE(8, 16, true) else KT_AGG_BM_CASE(8, 16, false) }
  else if (LT == 8) { if (keys) KT_AGG_BM_CASE(16, 8, true) else KT_AGG_BM_CASE(16, 8, false) }
  else { if (keys) KT_AGG_BM_CASE(16, 16, true) else KT_AGG_BM_CASE(16, 16, false) }
#endif
  if (after_scan) after_scan();
  const int max_words = (int)ix.bm_max_thr * partial_stride(pods.D);
  if (max_words > 0)
    hipLaunchKernelGGL(kt_reduce_bitmap_slabs, dim3((max_words + 63) / 64, ix.n_chunks), dim3(1024), 0, s, slab, ix.bm_chunks,
                       ix.bm_rank_t, nb, pods.D, sc.counts ? 1 : 0, sc.sign, partial);
  return ix.n_chunks == 1 ? "kt_aggregate_bitmap" : "kt_aggregate_bitmap_chunked";
}

}  // namespace kt
