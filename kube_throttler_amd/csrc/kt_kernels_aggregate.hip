// kt_kernels_aggregate.hip — kt_aggregate_bitmap + kt_reduce_bitmap_slabs: per-throttle `used` through the exact term bitmaps.
#include <cstdio>

#include "kt_scan.h"

namespace kt {

// LDS table of one chunk, 16-byte granules:
//   tv i64[n_thr][D] | tpres u32[n_thr] (request-key presence MASK) | tpods u32[n_thr]
//   counts mode (incremental engines): tv i64[n_thr][D] | tcnt u32[n_thr][D] (pods carrying the key) | tpods u32[n_thr]
__host__ __device__ inline uint32_t agg_tab_bytes(uint32_t n_thr, int D, bool counts) {
  const size_t per = (size_t)D * 8 + (counts ? (size_t)D * 4 + 4 : 8);
  return (uint32_t)(((size_t)n_thr * per + 15) & ~(size_t)15);
}

struct BmAggArgs {
  const uint64_t* meta;  // pod tables
  const uint16_t* latom;
  const int64_t* req;
  const uint32_t* lpair;  // raw labels: slow paths only
  const uint32_t* lkey;
  const SelProgram* sp;
  const uint32_t* slow_thr;
  unsigned long long* partial;  // target of the slow-list atomics (and of the slab reduction)
  unsigned char* slab;
  const int64_t* rows;  // nullable: the pods to scan are rows[0..n_rows) instead of row0 + [0, n_rows)
  int64_t row0, n_rows;
  BmIndexArgs ix;
  uint32_t off_rank, off_tab;
  uint32_t n_slow;
  int32_t D, DS, LS, T;
  int32_t counts;  // table keeps per-key pod counts instead of the presence mask
  int32_t sign;    // +1 / -1: the scanned pods are added to / removed from the target (delta scans)
};

static BmAggArgs make_bm_agg_args(const PodTable& pods, const AggScan& sc, const SelProgram& sp, const SelProgram* sp_dev,
                                  const IndexDev& ix, unsigned long long* partial, unsigned char* slab, uint32_t* total) {
  BmAggArgs a{};
  a.rows = sc.rows, a.row0 = sc.row0, a.n_rows = sc.n, a.counts = sc.counts ? 1 : 0, a.sign = sc.sign;
  a.meta = pods.meta, a.latom = pods.latom, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.sp = sp_dev, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow, a.partial = partial, a.slab = slab;
  a.D = pods.D, a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_rank = take(ix.bm_max_words * 64u * 2u);
  a.off_tab = take(agg_tab_bytes(ix.bm_max_thr, pods.D, sc.counts));
  plan_bitmap_index(ix, a.ix, take);
  *total = o;
  return a;
}

uint32_t aggregate_fixed_lds() { return 64; }

// kt_aggregate_bitmap — `used` partials of this GPU's pod rows: affectedPods + fold Add for all throttles
// (throttle_controller.go:116-119,221-246; clusterthrottle_controller.go:119-122,224-270).
// The workgroup walks the chunks of the index: chunk image in, table of the chunk's throttles zeroed, every tile of
// the workgroup scanned against it — wave-autonomous like kt_check_bitmap, lane = pod throughout: the lane holds its
// pod's request row in registers (loaded for counted pods only) and folds ResourceAmountOfPod into the LDS table
//   tv i64[n_thr][D] | tpres u32[n_thr] (request-key presence mask) | tpods u32[n_thr]
// for every term scan_tile reports (ds_add_u64 per non-zero dimension) — then the table is spilled to this
// (chunk, workgroup)'s slab; kt_reduce_bitmap_slabs sums the slabs into the partial buffer.
// No global atomics except for throttles with unconvertible selectors (the "slow" list).
template <int DT, int LA, bool VETO, int NEED>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_bitmap(const BmAggArgs a) {
  const int D = a.D, DS = a.DS;
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const int pstride = partial_stride(D);
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t n_rows = a.n_rows;
  const int64_t n_wtiles = (n_rows + kWave - 1) / kWave;
  const int64_t wstep = (int64_t)gridDim.x * (kBlockIx / kWave);
  KT_LDS const uint16_t* trank = (KT_LDS const uint16_t*)(lds + a.off_rank);
  const bool counts = a.counts != 0;
  for (uint32_t ci = 0; ci < a.ix.n_chunks; ++ci) {
    const BmChunk ch = a.ix.chunks[ci];
    const uint32_t n_thr = ch.n_thr;
    const uint32_t tab_bytes = agg_tab_bytes(n_thr, D, counts);
    lds_u64wp tv = (lds_u64wp)(lds + a.off_tab);
    lds_u32wp tpres = (lds_u32wp)(lds + a.off_tab + n_thr * (uint32_t)D * 8);  // mask [n] or counts [n][D]
    lds_u32wp tpods = tpres + (counts ? n_thr * (uint32_t)D : n_thr);
    __syncthreads();  // nobody reads the previous image / table any more
    for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
    lds_stage16((KT_LDS u32x4*)(lds + a.off_rank), (const u32x4*)(a.ix.blob + ch.img_off + ch.off_term_rank), ch.n_words * 8u);
    const BmView bm = open_chunk<VETO>(lds, a.ix, ch);
    __syncthreads();
    int64_t wt = (int64_t)blockIdx.x * (kBlockIx / kWave) + wave;
    for (; wt < n_wtiles; wt += wstep) {
      // ---- the tile's records, always from valid addresses (lanes past the end re-read the last row and are off)
      const int64_t i = wt * kWave + lane;
      const bool in = i < n_rows;
      const int64_t ic = min(i, n_rows - 1);
      const int64_t p = a.rows ? a.rows[ic] : a.row0 + ic;
      const uint64_t meta = a.meta[p];
      u32x4 raw[LA / 8];
      load_atoms<LA>(a.latom, p, raw);
      const uint32_t st = (uint32_t)(meta >> kMetaStateShift) & 0xFu;
      // shouldCountIn (throttle_controller.go:217-219); terminated pods are matched but not counted
      // (isNotFinished, pod_util.go:26-28) and only matter for error detection (slow list)
      const bool countable = in && (st & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
      const bool counted = countable && !(st & kPodFinished);
      if (__ballot(countable) == 0ull) continue;
      const uint32_t ns = countable ? (uint32_t)(meta & kMetaNsMask) : 0u;
      const uint32_t present = (uint32_t)(meta >> kMetaPresentShift) & 0xFFFFu;
      // ResourceAmountOfPod: the request row, for counted pods only (exec-masked 128-bit loads)
      int64_t v[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) v[d] = 0;
      if (counted) load_requests<DT>(a.req, DS, p, v);
      uint32_t ro[LA];
      atom_row_offsets<LA>(raw, bm.row_bytes, ro);

      // ---- throttles with unconvertible selectors have no rank: walked once (with the first chunk), straight to the
      //      result buffer
      if (ci == 0 && a.n_slow) {
        const SelProgram& sp = *a.sp;
        const uint32_t* lp = a.lpair + (uint64_t)p * (uint32_t)a.LS;
        const uint32_t* lk = a.lkey + (uint64_t)p * (uint32_t)a.LS;
        for (uint32_t ks = 0; ks < a.n_slow; ++ks) {
          const uint32_t t = a.slow_thr[ks];
          const uint32_t res = walk_slow_mem(sp, (int)t, sp.ns_term_ok + (size_t)ns * sp.gw, countable, lp, lk, a.LS);
          unsigned long long* pr = a.partial + (size_t)t * pstride;
          if (res & kSlowError) atomicAdd(pr + 2 * D + 1, (unsigned long long)(long long)a.sign);
          if ((res & kSlowMatched) && counted) {
            for (int d = 0; d < D; ++d)
              if ((present >> d) & 1u) {
                const int64_t vd = a.req[(uint64_t)p * (uint32_t)DS + d];
                if (vd != 0) atomicAdd(pr + d, (unsigned long long)(a.sign * vd));
                atomicAdd(pr + D + d, (unsigned long long)(long long)a.sign);
              }
            atomicAdd(pr + 2 * D, (unsigned long long)(long long)a.sign);
          }
        }
      }

      uint32_t last_r = 0xFFFFFFFFu;
      scan_tile<LA, VETO, NEED>(
          bm, counted, ns, ro,
          [&](bool has, uint32_t c) {
            const uint32_t tr = trank[c];
            const uint32_t r = tr & 0x7FFFu;  // chunk-local throttle rank
            // a throttle with several terms is counted once
            const bool ok = has && !((tr & kRankAdj) && r == last_r);
            if (ok) {
              last_r = r;
#pragma unroll
              for (int d = 0; d < DT; ++d)
                if (v[d] != 0) lds_add64(tv + r * (uint32_t)D + d, (unsigned long long)v[d]);  // padding dimensions hold 0
              if (counts) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
                  if ((present >> d) & 1u) lds_add(tpres + r * (uint32_t)D + d, 1u);
              } else {
                (void)__hip_atomic_fetch_or(tpres + r, present, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              lds_add(tpods + r, 1u);
            }
          },
          [&](uint32_t c) {
            return term_match_mem(*a.sp, bm.term_g[c], a.lpair + (uint64_t)p * (uint32_t)a.LS, a.lkey + (uint64_t)p * (uint32_t)a.LS, a.LS);
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

#define KT_AGG_BM_CASE(DT_, LA_, VETO_, NEED_)                                                                \
  {                                                                                                           \
    auto kfn = kt_aggregate_bitmap<DT_, LA_, VETO_, NEED_>;                                                   \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bm);     \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bm, s, bm_args);                                                      \
  }

static inline int agg_blocks(int64_t n_rows) {
  int64_t b = (n_rows + kBlockIx - 1) / kBlockIx;
  return (int)(b < 1 ? 1 : b > kCUs ? kCUs : b);
}

// `partial` must be zeroed by the caller.  Returns the dispatched scan kernel's symbol, nullptr when a chunk of the
// index does not fit the workgroup's LDS.
const char* launch_aggregate_indexed(const PodTable& pods, const AggScan& sc, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, unsigned long long* partial, void* slab_, hipStream_t s,
                              const std::function<void()>& after_scan) {
  const int64_t n_rows = sc.n;
  if (n_rows <= 0 || sp.T <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LA = pods.LA;
  unsigned char* slab = (unsigned char*)slab_;
  const int nb = agg_blocks(n_rows);
  dim3 g_(nb), b_(kBlockIx);
  uint32_t bm_total = 0;
  const BmAggArgs bm_args = make_bm_agg_args(pods, sc, sp, sp_dev, ix, partial, slab, &bm_total);
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  const size_t lds_bm = bm_total;
  static const bool dbg_lds = getenv("KT_DEBUG_LDS") != nullptr;
  if (dbg_lds) fprintf(stderr, "kt_aggregate_bitmap: lds=%u chunks=%u largest LDS part=%u max thr=%u T=%d\n", bm_total, ix.n_chunks, ix.bm_max_lds, ix.bm_max_thr, sp.T);
#ifdef KT_FAST_BUILD
  KT_AGG_BM_CASE(8, 8, false, 2)
#else
  if (!ix.rich) { if (DT <= 8) KT_AGG_BM_CASE(8, 8, false, 2) else KT_AGG_BM_CASE(16, 8, false, 2) }
  else if (LA <= 16) { if (DT <= 8) KT_AGG_BM_CASE(8, 16, true, 3) else KT_AGG_BM_CASE(16, 16, true, 3) }
  else { if (DT <= 8) KT_AGG_BM_CASE(8, 32, true, 3) else KT_AGG_BM_CASE(16, 32, true, 3) }
#endif
  if (after_scan) after_scan();
  const int max_words = (int)ix.bm_max_thr * partial_stride(pods.D);
  if (max_words > 0)
    hipLaunchKernelGGL(kt_reduce_bitmap_slabs, dim3((max_words + 63) / 64, ix.n_chunks), dim3(1024), 0, s, slab, ix.bm_chunks,
                       ix.bm_rank_t, nb, pods.D, sc.counts ? 1 : 0, sc.sign, partial);
  return ix.n_chunks == 1 ? "kt_aggregate_bitmap" : "kt_aggregate_bitmap_chunked";
}

}  // namespace kt
