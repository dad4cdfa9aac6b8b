// kt_kernels_aggregate.hip — kt_aggregate_indexed + kt_reduce_partials: per-throttle `used` through the index.
#include "kt_bitmap_scan.h"

namespace kt {

// ---------------------------------------------------------------------------------------------------
// kt_aggregate_indexed — affectedPods + fold Add for all throttles (throttle_controller.go:116-119,
// 221-246; clusterthrottle_controller.go:119-122,224-270) through the index.
// LDS table layout per workgroup: int64 v[T][D] | uint32 cnt[T][D+2]  (presence counts, pods, errors).
// ---------------------------------------------------------------------------------------------------
// mode 1: counts are uint32 [T][D+2]; mode 2 (index also in LDS): counts are uint16 packed two per word
__host__ __device__ inline size_t lds_table_bytes(int T, int D, bool cnt16) {
  const size_t cnt = (size_t)T * (D + 2);
  return (size_t)T * D * 8 + (cnt16 ? ((cnt + 1) / 2) * 4 : cnt * 4);
}

// AGG_MODE 0: no LDS table (global atomics), index through L2
//          1: LDS table (u32 counts), index through L2
//          2: LDS table (packed u16 counts) AND hash slots + postings staged in LDS
template <int DT, int LT, bool KEYS, int AGG_MODE>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_indexed(PodTable pods, int64_t n_rows, SelProgram sp,
                                                                IndexDev ix, unsigned long long* partial,
                                                                unsigned char* slab, uint32_t q_cap) {
  constexpr bool LDSTAB = AGG_MODE != 0, CNT16 = AGG_MODE == 2, LDSIX = AGG_MODE == 2;
  const int D = pods.D, stride = partial_stride(D), T = sp.T;
  // LDS carve: [queue u32 x q_cap][q_count + pad][table: int64 v[T][D] | counts][index copy (mode 2)]
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  lds_u32wp q = (lds_u32wp)lds;
  lds_u32wp q_count = q + q_cap;
  const uint32_t tab_off = q_cap * 4 + 16;
  const uint32_t tab_bytes = LDSTAB ? (uint32_t)((lds_table_bytes(T, D, CNT16) + 15) & ~(size_t)15) : 0u;
  lds_u64wp tv = (lds_u64wp)(lds + tab_off);
  lds_u32wp tc = (lds_u32wp)(lds + tab_off + (uint32_t)T * D * 8);
  lds_u4p l_slots = (lds_u4p)(lds + tab_off + tab_bytes);
  lds_u4p l_posts = l_slots + ix.n_slots;
  if (LDSTAB) {
    for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + tab_off))[i] = 0u;
  }
  if (LDSIX) {
    KT_LDS u32x4* dst = (KT_LDS u32x4*)(lds + tab_off + tab_bytes);
    const u32x4* src_s = (const u32x4*)ix.slots;
    const u32x4* src_p = (const u32x4*)ix.postings;
    const uint32_t ns16 = ix.n_slots, np16 = ix.n_postings * 2;
    for (uint32_t i = threadIdx.x; i < ns16; i += kBlockIx) dst[i] = src_s[i];
    for (uint32_t i = threadIdx.x; i < np16; i += kBlockIx) dst[ns16 + i] = src_p[i];
  }
  auto cnt_add = [&](uint32_t t, uint32_t j) {  // counts[t][j] += 1   (j < D: key presence, D: pods, D+1: errors)
    const uint32_t idx = t * (uint32_t)(D + 2) + j;
    if (CNT16) lds_add(tc + (idx >> 1), 1u << ((idx & 1u) * 16u));
    else lds_add(tc + idx, 1u);
  };
  auto add_pod = [&](uint32_t t, int64_t p) {  // used[t] += ResourceAmountOfPod(p): lane-serial form (overflow path)
    const uint32_t present = pods.flags[p] >> kPresentShift;
#pragma unroll
    for (int d = 0; d < DT; ++d)
      if (d < D && ((present >> d) & 1u)) {
        const int64_t v = pods.req[(int64_t)p * pods.DS + d];
        if (LDSTAB) {
          if (v != 0) lds_add64(tv + t * (uint32_t)D + d, (unsigned long long)v);
          cnt_add(t, (uint32_t)d);
        } else {
          if (v != 0) atomicAdd(partial + (size_t)t * stride + d, (unsigned long long)v);
          atomicAdd(partial + (size_t)t * stride + D + d, 1ull);
        }
      }
    if (LDSTAB) cnt_add(t, (uint32_t)D);
    else atomicAdd(partial + (size_t)t * stride + 2 * D, 1ull);
  };
  const int64_t n_tiles = (n_rows + kBlockIx - 1) / kBlockIx;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t p = tile * kBlockIx + threadIdx.x;
    if (threadIdx.x == 0) *q_count = 0u;
    __syncthreads();
    // ---- phase 1: lane = pod: enumerate matches of counted pods
    const uint32_t fl = p < n_rows ? pods.flags[p] : 0u;
    // shouldCountIn (throttle_controller.go:217-219)
    const bool countable = (fl & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
    const bool not_finished = !(fl & kPodFinished);  // isNotFinished (pod_util.go:26-28)
    if (countable && (not_finished || ix.n_slow != 0)) {  // terminated pods only matter for error detection
      uint32_t lp[LT], lk[LT];
      const uint32_t ns = pods.ns[p];
      load_labels<LT, KEYS>(pods.lpair, pods.lkey, pods.LS, p, lp, lk);
      auto on_match = [&](uint32_t t) {
        if (!not_finished) return;  // matched but not counted
        // wave-aggregated push (one LDS atomic per wave); a full queue folds the pod in directly
        const uint64_t mask = __ballot(true);
        const uint32_t lane = __lane_id();
        const uint32_t leader = (uint32_t)__ffsll((unsigned long long)mask) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = lds_add(q_count, (uint32_t)__popcll(mask));
        base = __shfl(base, leader) + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (base < q_cap) q[base] = (uint32_t)threadIdx.x << 20 | t;
        else add_pod(t, p);
      };
      if (not_finished) {
        if (LDSIX) enumerate_matches<LT, KEYS>(sp, ix, l_slots, l_posts, ns, lp, lk, on_match);
        else enumerate_matches<LT, KEYS>(sp, ix, (const u32x4*)ix.slots, (const u32x4*)ix.postings, ns, lp, lk, on_match);
      }
      const uint32_t* ns_row = sp.ns_term_ok + (size_t)ns * sp.gw;
      for (uint32_t k = 0; k < ix.n_slow; ++k) {
        bool matched, err;
        const int t = (int)ix.slow_thr[k];
        walk_slow<LT, KEYS>(sp, t, ns_row, true, lp, lk, matched, err);
        if (err) {
          if (LDSTAB) cnt_add((uint32_t)t, (uint32_t)D + 1u);
          else atomicAdd(partial + (size_t)t * stride + 2 * D + 1, 1ull);
        }
        if (matched) on_match((uint32_t)t);
      }
    }
    __syncthreads();
    // ---- phase 2: lane = (match, dimension): fold the pod's amount into the table
    const uint32_t qn = min(*q_count, q_cap);
    {
      constexpr int MPW = kWave / DT;
      const uint32_t lane = threadIdx.x & (kWave - 1), d = lane % DT, ml = lane / DT;
      const uint32_t wave = threadIdx.x / kWave;
      for (uint32_t base = wave * MPW; base < qn; base += (kBlockIx / kWave) * MPW) {
        const uint32_t j = base + ml;
        if (j >= qn) continue;
        const uint32_t e = q[j];
        const uint32_t t = e & 0xFFFFFu;
        const int64_t mp = tile * kBlockIx + (e >> 20);
        const uint32_t present = pods.flags[mp] >> kPresentShift;
        if ((int)d < D && ((present >> d) & 1u)) {
          const int64_t v = pods.req[(int64_t)mp * pods.DS + d];
          if (LDSTAB) {
            if (v != 0) lds_add64(tv + t * (uint32_t)D + d, (unsigned long long)v);
            cnt_add(t, d);
          } else {
            if (v != 0) atomicAdd(partial + (size_t)t * stride + d, (unsigned long long)v);
            atomicAdd(partial + (size_t)t * stride + D + d, 1ull);
          }
        }
        if (d == 0) {
          if (LDSTAB) cnt_add(t, (uint32_t)D);
          else atomicAdd(partial + (size_t)t * stride + 2 * D, 1ull);
        }
      }
    }
    __syncthreads();
  }
  if (LDSTAB) {  // spill this workgroup's table (coalesced 16-byte stores); kt_reduce_partials sums the slabs
    __syncthreads();
    u32x4* dst = (u32x4*)(slab + (size_t)blockIdx.x * tab_bytes);
    lds_u4p src = (lds_u4p)(lds + tab_off);
    for (uint32_t i = threadIdx.x; i < tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_aggregate_bitmap — kt_aggregate_indexed for selector programs whose bitmap form fits in LDS next to the
// per-workgroup `used` table.  WAVE-AUTONOMOUS like kt_check_bitmap: after staging, every wave walks its own
// 64-pod tiles without workgroup barriers; phase 1 parks matches in private LDS columns, a convergent
// compaction makes them dense, phase 2 folds (match, dimension) lanes into the workgroup's LDS table:
//     int64 v[T][D] (ds_add_u64) | uint32 present[T] (ds_or_b32 of the pod's key mask) | uint32 pods[T]
// Key presence travels as a mask here and is expanded to 0/1 contributor counts by kt_reduce_bitmap_slabs
// (sum > 0 <=> some workgroup saw the key), so the all-reduced buffer keeps the [T][2D+2] layout.
// ---------------------------------------------------------------------------------------------------
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
                                  const IndexDev& ix, unsigned long long* partial, unsigned char* slab, uint32_t* total) {
  BmAggArgs a{};
  a.ns = pods.ns, a.flags = pods.flags, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.sp = sp_dev, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow, a.partial = partial, a.slab = slab, a.n_rows = n_rows;
  a.D = pods.D, a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_list = take((kBlockIx / kWave) * kAggListCap * 4);
  a.off_pres = take(kBlockIx * 4);
  a.tab_bytes = agg_bitmap_tab_bytes(sp.T, pods.D);
  a.off_tab = take(a.tab_bytes);
  plan_bitmap_index(ix, a.ix, take);
  *total = o;
  return a;
}

// kt_aggregate_bitmap — `used` partials of this GPU's pod rows (reconcile aggregation,
// throttle_controller.go:116-133) for selector programs whose bitmap form fits in LDS.
// Wave-autonomous like kt_check_bitmap: lane = pod finds the tile's (pod, throttle) matches
// (bitmap_scan_tile), lane = (match, dimension pair) folds ResourceAmountOfPod into the workgroup's LDS
// table  tv i64[T][D] | tpres u32[T] (request-key presence mask) | tpods u32[T];  the table is spilled to
// this workgroup's slab at the end and kt_reduce_bitmap_slabs sums the slabs.
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_bitmap(const BmAggArgs a) {
  const int D = a.D, DS = a.DS, T = a.T;
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  lds_u64wp tv = (lds_u64wp)(lds + a.off_tab);
  lds_u32wp tpres = (lds_u32wp)(lds + a.off_tab + (uint32_t)T * D * 8);
  lds_u32wp tpods = tpres + T;
  for (uint32_t i = threadIdx.x; i < a.tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
  const BmView bm = stage_bitmap_index(lds, a.ix);
  __syncthreads();
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  lds_u32wp list = (lds_u32wp)(lds + a.off_list) + wave * kAggListCap;
  lds_u32wp l_pres = (lds_u32wp)(lds + a.off_pres) + wave * kWave;  // [64] request-key presence masks of the tile
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
    l_pres[lane] = fl >> kPresentShift;

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
          if (c.x.x != 0) lds_add64(tv + c.t * (uint32_t)D + 2 * dp, (unsigned long long)c.x.x);
          if (c.x.y != 0) lds_add64(tv + c.t * (uint32_t)D + 2 * dp + 1, (unsigned long long)c.x.y);  // padding dimension is 0
          if (dp == 0) {
            (void)__hip_atomic_fetch_or(tpres + c.t, c.pres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_add(tpods + c.t, 1u);
          }
        }
        c = nx;
      }
    };
    bitmap_scan_tile<LT, KEYS, kAggListCap>(bm, a.sp, a.slow_thr, a.n_slow, countable && not_finished, countable, ns, cur.lp,
                                            cur.lk, list, lane, drain, [&](uint32_t t) {  // rare: straight to the result buffer
                                              atomicAdd(a.partial + (size_t)t * partial_stride(D) + 2 * D + 1, 1ull);
                                            });
    cur = nxt;
  }
  __syncthreads();  // spill this workgroup's table (coalesced 16-byte stores); kt_reduce_bitmap_slabs sums the slabs
  u32x4* dst = (u32x4*)(a.slab + (size_t)blockIdx.x * a.tab_bytes);
  lds_u4p src = (lds_u4p)(lds + a.off_tab);
  for (uint32_t i = threadIdx.x; i < a.tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
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
    auto kfn = kt_aggregate_bitmap<DT_, LT_, KEYS_>;                                                          \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bm);     \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bm, s, bm_args);                                                      \
  }

// partial[t][j] = sum over workgroup slabs (j < D: values; D <= j < 2D+2: counts).
// 64 output words per workgroup x 4 slab groups: every thread streams n_slabs/4 independent loads.
__global__ __launch_bounds__(256) void kt_reduce_partials(const unsigned char* slab, int n_slabs, int T, int D, int cnt16,
                                                         unsigned long long* partial) {
  __shared__ unsigned long long part[4][64];
  const int stride = partial_stride(D);
  const size_t pitch = (lds_table_bytes(T, D, cnt16 != 0) + 15) & ~(size_t)15;
  const int words = T * stride;
  const int wl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + wl;
  unsigned long long acc = 0;
  if (w < words) {
    const int t = w / stride, j = w - t * stride;
    if (j < D) {
      const unsigned char* base = slab + ((size_t)t * D + j) * 8;
#pragma unroll 8
      for (int b = g; b < n_slabs; b += 4) acc += *(const unsigned long long*)(base + b * pitch);
    } else {
      const size_t idx = (size_t)t * (D + 2) + (j - D);
      const unsigned char* base = slab + (size_t)T * D * 8 + idx * (cnt16 ? 2 : 4);
      if (cnt16) {
#pragma unroll 8
        for (int b = g; b < n_slabs; b += 4) acc += *(const unsigned short*)(base + b * pitch);
      } else {
#pragma unroll 8
        for (int b = g; b < n_slabs; b += 4) acc += *(const unsigned int*)(base + b * pitch);
      }
    }
  }
  part[g][wl] = acc;
  __syncthreads();
  if (g == 0 && w < words) partial[w] = part[0][wl] + part[1][wl] + part[2][wl] + part[3][wl];
}

static inline int agg_blocks(int64_t n_rows) {
  int64_t b = (n_rows + kBlockIx - 1) / kBlockIx;
  return (int)(b < 1 ? 1 : b > kCUs ? kCUs : b);
}

size_t aggregate_slab_bytes(int T, int D) {
  const size_t bytes = lds_table_bytes(T, D, false);
  if (bytes + 2048 * 4 + 16 > (size_t)kMaxLds) return 0;
  return (size_t)kCUs * ((bytes + 15) & ~(size_t)15);
}

const char* launch_aggregate_indexed(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, bool keys, unsigned long long* partial, void* slab_, hipStream_t s,
                              const std::function<void()>& after_scan) {
  if (n_rows <= 0 || sp.T <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  unsigned char* slab = (unsigned char*)slab_;
  const int nb = agg_blocks(n_rows);
  const size_t ix_bytes = (size_t)ix.n_slots * sizeof(IndexSlot) + (size_t)ix.n_postings * sizeof(Posting);
  const size_t tab32 = (lds_table_bytes(sp.T, pods.D, false) + 15) & ~(size_t)15;
  const size_t tab16 = (lds_table_bytes(sp.T, pods.D, true) + 15) & ~(size_t)15;
  const int64_t pods_per_block = ((n_rows + kBlockIx - 1) / kBlockIx + nb - 1) / nb * kBlockIx;
  // mode 2: table (u16 counts) + index + a short queue, all in LDS; mode 1: table only; mode 0: global atomics
  int mode = 0;
  uint32_t q_cap = kQueueCap;
  size_t tab = 0;
  if (slab != nullptr) {
    if (pods_per_block <= 65535 && 2048 * 4 + 16 + tab16 + ix_bytes <= (size_t)kMaxLds) mode = 2, q_cap = 2048, tab = tab16;
    else if (kQueueCap * 4 + 16 + tab32 <= (size_t)kMaxLds) mode = 1, tab = tab32;
    else if (2048 * 4 + 16 + tab32 <= (size_t)kMaxLds) mode = 1, q_cap = 2048, tab = tab32;
  }
  dim3 g_(nb), b_(kBlockIx);
  // small-T regime: LDS table + the whole selector program as LDS-resident bitmaps
  if (slab != nullptr && ix.bm_words != 0) {
    uint32_t bm_total = 0;
    const BmAggArgs bm_args = make_bm_agg_args(pods, n_rows, sp, sp_dev, ix, partial, slab, &bm_total);
    if (bm_total <= (uint32_t)kMaxLds) {
      const size_t lds_bm = bm_total;
#ifdef KT_FAST_BUILD
      KT_AGG_BM_CASE(8, 8, false)
#else
      if (DT <= 8 && LT == 8) { if (keys) KT_AGG_BM_CASE(8, 8, true) else KT_AGG_BM_CASE(8, 8, false) }
      else if (DT <= 8) { if (keys) KT_AGG_BM_CASE(8, 16, true) else KT_AGG_BM_CASE(8, 16, false) }
      else if (LT == 8) { if (keys) KT_AGG_BM_CASE(16, 8, true) else KT_AGG_BM_CASE(16, 8, false) }
      else { if (keys) KT_AGG_BM_CASE(16, 16, true) else KT_AGG_BM_CASE(16, 16, false) }
#endif
      const int words = sp.T * partial_stride(pods.D);
      if (after_scan) after_scan();
      hipLaunchKernelGGL(kt_reduce_bitmap_slabs, dim3((words + 63) / 64), dim3(1024), 0, s, slab, nb, sp.T, pods.D, partial);
      return "kt_aggregate_bitmap";
    }
  }
  const size_t lds_bytes = q_cap * 4 + 16 + tab + (mode == 2 ? ix_bytes : 0);
#define KT_IX_ARGS pods, n_rows, sp, ix, partial, slab, q_cap
  if (mode == 2) KT_IX_DISPATCH2(kt_aggregate_indexed, DT, LT, keys, 2);
  else if (mode == 1) KT_IX_DISPATCH2(kt_aggregate_indexed, DT, LT, keys, 1);
  else KT_IX_DISPATCH2(kt_aggregate_indexed, DT, LT, keys, 0);
#undef KT_IX_ARGS
  if (after_scan) after_scan();
  if (mode != 0) {
    const int words = sp.T * partial_stride(pods.D);
    hipLaunchKernelGGL(kt_reduce_partials, dim3((words + 63) / 64), dim3(256), 0, s, slab, nb, sp.T, pods.D,
                       mode == 2 ? 1 : 0, partial);
  }
  return "kt_aggregate_indexed";
}


}  // namespace kt
