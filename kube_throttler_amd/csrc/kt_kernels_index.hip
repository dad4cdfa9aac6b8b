// kt_kernels_index.hip — indexed pod x throttle scans for gfx950: work ~ (pods + candidate terms).
//
// lane = pod.  Each lane probes the label-atom hash index (kt_index.h) with its own labels, verifies the
// few candidate terms it finds, and classifies / accumulates only true matches.  Decisions still cover
// the full P x T matrix: every pair not enumerated is "not affected" by construction of the index.
#include "kt_index.h"
#include "kt_kernels_common.h"
#include "kt_launch.h"

namespace kt {

constexpr int kBlockIx = 256;

static inline int grid_ix(int64_t n) {
  int64_t b = (n + kBlockIx - 1) / kBlockIx;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (int)b;
}

template <int LT, bool KEYS>
struct Matcher {
  const SelProgram& sp;
  const uint32_t* ns_row;
  const uint32_t (&lp)[LT];
  const uint32_t (&lk)[LT];

  __device__ __forceinline__ bool ns_ok(uint32_t g) const { return (ns_row[g >> 5] >> (g & 31)) & 1u; }

  // term g matches this pod AND no earlier term of the same throttle does (so a throttle whose
  // selector has several matching terms is reported once — by its first matching term).
  __device__ __forceinline__ bool owns_match(uint32_t g, bool check_ns, uint32_t& t_out) const {
    if (check_ns && !ns_ok(g)) return false;
    if (!term_match<LT, KEYS>(sp, g, lp, lk)) return false;
    const uint32_t t = sp.term_thr[g];
    for (uint32_t g2 = sp.thr_term_off[t]; g2 < g; ++g2)
      if (ns_ok(g2) && term_match<LT, KEYS>(sp, g2, lp, lk)) return false;
    t_out = t;
    return true;
  }
};

// One hash lookup: (begin, count) of the posting list filed under `key` (count 0 when absent).
__device__ __forceinline__ uint2 lookup(const IndexDev& ix, uint64_t key) {
  uint32_t h = index_hash(key, ix.mask);
  for (;;) {
    const IndexSlot s = ix.slots[h];
    if (s.key == key) return make_uint2(s.begin, s.count);
    if (s.key == 0) return make_uint2(0u, 0u);
    h = (h + 1) & ix.mask;
  }
}

template <int LT, bool KEYS, class F>
__device__ __forceinline__ void enumerate_matches(const SelProgram& sp, const IndexDev& ix, uint32_t ns,
                                                  const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], F&& on_match) {
  const Matcher<LT, KEYS> m{sp, sp.ns_term_ok + (size_t)ns * sp.gw, lp, lk};
  const uint64_t scope = (uint64_t)(ns + 1) << 32;
  // all lookups first (independent loads in flight together), then the posting walks
  uint2 rn[LT], rc[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const uint32_t pair = lp[l];
    rn[l] = pair ? lookup(ix, scope | pair) : make_uint2(0u, 0u);          // Throttles of the pod's namespace
    rc[l] = pair ? lookup(ix, (uint64_t)pair) : make_uint2(0u, 0u);        // ClusterThrottles
  }
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    for (uint32_t k = 0; k < rn[l].y; ++k) {
      uint32_t t;
      if (m.owns_match(ix.postings[rn[l].x + k], false, t)) on_match(t);
    }
    for (uint32_t k = 0; k < rc[l].y; ++k) {
      uint32_t t;
      if (m.owns_match(ix.postings[rc[l].x + k], true, t)) on_match(t);
    }
  }
  if (KEYS && ix.has_key_atoms) {
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if (lk[l] == 0) continue;
      const uint32_t ka = kKeyAtom | lk[l];
      const uint2 a = lookup(ix, scope | ka), b = lookup(ix, (uint64_t)ka);
      for (uint32_t k = 0; k < a.y; ++k) {
        uint32_t t;
        if (m.owns_match(ix.postings[a.x + k], false, t)) on_match(t);
      }
      for (uint32_t k = 0; k < b.y; ++k) {
        uint32_t t;
        if (m.owns_match(ix.postings[b.x + k], true, t)) on_match(t);
      }
    }
  }
  for (uint32_t k = ix.uni_ns_off[ns]; k < ix.uni_ns_off[ns + 1]; ++k) {
    uint32_t t;
    if (m.owns_match(ix.uni_ns[k], false, t)) on_match(t);
  }
  for (uint32_t k = 0; k < ix.n_uni_cluster; ++k) {
    uint32_t t;
    if (m.owns_match(ix.uni_cluster[k], true, t)) on_match(t);
  }
}

// Throttles with an unconvertible podSelector term: in-order walk, error when the bad term is reached
// before a match (same routine as the dense kernels; t is wave-uniform).
template <int LT, bool KEYS>
__device__ __forceinline__ void walk_slow(const SelProgram& sp, int t, const uint32_t* ns_row, bool lane_on,
                                          const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], bool& matched, bool& err) {
  matched = false;
  err = false;
  bool open = lane_on;
  const uint32_t g1 = sp.thr_term_off[t + 1];
  for (uint32_t g = sp.thr_term_off[t]; g < g1; ++g) {
    const bool applies = open && ((ns_row[g >> 5] >> (g & 31)) & 1u);
    if (sp.term_flags[g] & kTermPodSelInvalid) {
      err |= applies;
      open &= !applies;
      continue;
    }
    const bool mt = applies && term_match<LT, KEYS>(sp, g, lp, lk);
    matched |= mt;
    open &= !mt;
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_check_indexed — PreFilter for n pods (plugin.go:148-215) through the index.
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_check_indexed(PodTable pods, int64_t n, const int64_t* rows,
                                                            SelProgram sp, IndexDev ix, const void* recs_,
                                                            uint64_t* summary, uint8_t* status) {
  const CheckRec<DT>* recs = (const CheckRec<DT>*)recs_;
  for (int64_t i = (int64_t)blockIdx.x * kBlockIx + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlockIx) {
    const int64_t p = rows ? rows[i] : i;
    const uint32_t fl = pods.flags[p];
    if (!(fl & kPodValid)) {
      summary[i] = 0;
      continue;
    }
    PodRegs<DT, LT, KEYS> r;
    load_pod<DT, LT, KEYS>(pods, p, r, true);
    // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
    bool pod_err = !sp.ns_valid[r.ns];
    uint32_t n_exc = 0, n_act = 0, n_ins = 0;
    uint8_t* srow = status ? status + i * sp.T : nullptr;
    auto on_match = [&](uint32_t t) {
      const uint32_t st = classify<DT>(recs + t, r.v, r.nzmask);
      n_exc += st == 4u;
      n_act += st == 2u;
      n_ins += st == 3u;
      if (srow) srow[t] = (uint8_t)st;
    };
    enumerate_matches<LT, KEYS>(sp, ix, r.ns, r.lp, r.lk, on_match);
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)r.ns * sp.gw;
    for (uint32_t k = 0; k < ix.n_slow; ++k) {
      bool matched, err;
      const int t = (int)ix.slow_thr[k];
      walk_slow<LT, KEYS>(sp, t, ns_row, true, r.lp, r.lk, matched, err);
      pod_err |= err;
      if (matched) on_match((uint32_t)t);
    }
    summary[i] = pack_summary(n_exc, n_act, n_ins, pod_err);
    if (srow && pod_err)
      for (int t = 0; t < sp.T; ++t) srow[t] = 255;
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_aggregate_indexed — affectedPods + fold Add for all throttles (throttle_controller.go:116-119,
// 221-246; clusterthrottle_controller.go:119-122,224-270) through the index.
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_indexed(PodTable pods, int64_t n_rows, SelProgram sp,
                                                                IndexDev ix, unsigned long long* partial) {
  const int D = pods.D, stride = partial_stride(D);
  for (int64_t p = (int64_t)blockIdx.x * kBlockIx + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * kBlockIx) {
    const uint32_t fl = pods.flags[p];
    // shouldCountIn (throttle_controller.go:217-219)
    if ((fl & (kPodValid | kPodSchedMatch | kPodScheduled)) != (kPodValid | kPodSchedMatch | kPodScheduled)) continue;
    const bool not_finished = !(fl & kPodFinished);
    PodRegs<DT, LT, KEYS> r;
    load_pod<DT, LT, KEYS>(pods, p, r, not_finished);
    const uint32_t present = fl >> kPresentShift;
    auto on_match = [&](uint32_t t) {
      if (!not_finished) return;  // terminated pods are matched but not counted (isNotFinished, pod_util.go:26-28)
      unsigned long long* row = partial + (size_t)t * stride;
#pragma unroll
      for (int d = 0; d < DT; ++d)
        if (d < D && ((present >> d) & 1u)) {
          if (r.v[d] != 0) atomicAdd(row + d, (unsigned long long)r.v[d]);
          atomicAdd(row + D + d, 1ull);
        }
      atomicAdd(row + 2 * D, 1ull);
    };
    if (not_finished) enumerate_matches<LT, KEYS>(sp, ix, r.ns, r.lp, r.lk, on_match);
    const uint32_t* ns_row = sp.ns_term_ok + (size_t)r.ns * sp.gw;
    for (uint32_t k = 0; k < ix.n_slow; ++k) {
      bool matched, err;
      const int t = (int)ix.slow_thr[k];
      walk_slow<LT, KEYS>(sp, t, ns_row, true, r.lp, r.lk, matched, err);
      if (err) atomicAdd(partial + (size_t)t * stride + 2 * D + 1, 1ull);
      if (matched) on_match((uint32_t)t);
    }
  }
}

#define KT_IX_DISPATCH(NAME, DT_, LT_, KEYS_, GRID, STREAM, ...)                                                 \
  do {                                                                                                           \
    dim3 g_(GRID), b_(kBlockIx);                                                                                 \
    if (DT_ == 4 && LT_ == 8 && !KEYS_) hipLaunchKernelGGL((NAME<4, 8, false>), g_, b_, 0, STREAM, __VA_ARGS__);   \
    else if (DT_ == 4 && LT_ == 8) hipLaunchKernelGGL((NAME<4, 8, true>), g_, b_, 0, STREAM, __VA_ARGS__);         \
    else if (DT_ == 4 && !KEYS_) hipLaunchKernelGGL((NAME<4, 16, false>), g_, b_, 0, STREAM, __VA_ARGS__);         \
    else if (DT_ == 4) hipLaunchKernelGGL((NAME<4, 16, true>), g_, b_, 0, STREAM, __VA_ARGS__);                    \
    else if (DT_ == 8 && LT_ == 8 && !KEYS_) hipLaunchKernelGGL((NAME<8, 8, false>), g_, b_, 0, STREAM, __VA_ARGS__); \
    else if (DT_ == 8 && LT_ == 8) hipLaunchKernelGGL((NAME<8, 8, true>), g_, b_, 0, STREAM, __VA_ARGS__);         \
    else if (DT_ == 8 && !KEYS_) hipLaunchKernelGGL((NAME<8, 16, false>), g_, b_, 0, STREAM, __VA_ARGS__);         \
    else if (DT_ == 8) hipLaunchKernelGGL((NAME<8, 16, true>), g_, b_, 0, STREAM, __VA_ARGS__);                    \
    else if (LT_ == 8 && !KEYS_) hipLaunchKernelGGL((NAME<16, 8, false>), g_, b_, 0, STREAM, __VA_ARGS__);         \
    else if (LT_ == 8) hipLaunchKernelGGL((NAME<16, 8, true>), g_, b_, 0, STREAM, __VA_ARGS__);                    \
    else if (!KEYS_) hipLaunchKernelGGL((NAME<16, 16, false>), g_, b_, 0, STREAM, __VA_ARGS__);                    \
    else hipLaunchKernelGGL((NAME<16, 16, true>), g_, b_, 0, STREAM, __VA_ARGS__);                                 \
  } while (0)

void launch_aggregate_indexed(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const IndexDev& ix,
                              bool keys, unsigned long long* partial, hipStream_t s) {
  if (n_rows <= 0 || sp.T <= 0) return;
  const int DT = dt_bucket(pods.D), LT = lt_bucket(pods.L);
  KT_IX_DISPATCH(kt_aggregate_indexed, DT, LT, keys, grid_ix(n_rows), s, pods, n_rows, sp, ix, partial);
}

void launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const IndexDev& ix, bool keys, const void* recs, uint64_t* summary, uint8_t* status,
                          hipStream_t s) {
  if (n <= 0) return;
  const int DT = dt_bucket(pods.D), LT = lt_bucket(pods.L);
  if (status) (void)hipMemsetAsync(status, 0, (size_t)n * (size_t)sp.T, s);
  KT_IX_DISPATCH(kt_check_indexed, DT, LT, keys, grid_ix(n), s, pods, n, rows_dev, sp, ix, recs, summary, status);
}

}  // namespace kt
