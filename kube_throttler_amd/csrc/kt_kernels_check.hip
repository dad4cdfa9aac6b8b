// kt_kernels_check.hip — kt_check_indexed: PreFilter for n pods through the label-atom index (gfx950).
#include "kt_index_device.h"

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
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        lp[l] = l < pods.L ? pods.lpair[(int64_t)l * pods.cap + p] : 0u;
        lk[l] = (KEYS && l < pods.L) ? pods.lkey[(int64_t)l * pods.cap + p] : 0u;
      }
      // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
      pod_err = !sp.ns_valid[ns];
      auto on_match = [&](uint32_t t) {
        if (push_match(true, (uint32_t)threadIdx.x << 20 | t, q, q_count)) return;
        // queue full: classify inline (rare)
        int64_t v[DT];
        uint32_t nz = 0;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          v[d] = d < pods.D ? pods.req[(int64_t)p * pods.D + d] : 0;
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
        const int64_t v = (valid && (int)d < pods.D) ? pods.req[(int64_t)mp * pods.D + d] : 0;
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

void launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const IndexDev& ix, bool keys, const void* recs, uint64_t* summary, uint8_t* status,
                          hipStream_t s) {
  if (n <= 0) return;
  const int DT = dt_bucket_ix(pods.D), LT = lt_bucket(pods.L);
  if (status) (void)hipMemsetAsync(status, 0, (size_t)n * (size_t)sp.T, s);
  int64_t nb = (n + kBlockIx - 1) / kBlockIx;
  if (nb > kCUs) nb = kCUs;
  dim3 g_((unsigned)nb), b_(kBlockIx);
  static const int dbg = getenv("KT_DEBUG_MODE") ? atoi(getenv("KT_DEBUG_MODE")) : 0;
  const size_t ix_bytes = (size_t)ix.n_slots * sizeof(IndexSlot) + (size_t)ix.n_postings * sizeof(Posting);
  const size_t fixed_bytes = kBlockIx * 8 + kQueueCap * 4 + 16;
  const bool lds_ix = ix_bytes + fixed_bytes <= (size_t)kMaxLds && n >= 4 * kBlockIx;
  const size_t lds_bytes = fixed_bytes + (lds_ix ? ((ix_bytes + 15) & ~(size_t)15) : 0);
#define KT_IX_ARGS pods, n, rows_dev, sp, ix, recs, summary, status, dbg
  if (lds_ix) KT_IX_DISPATCH2(kt_check_indexed, DT, LT, keys, true);
  else KT_IX_DISPATCH2(kt_check_indexed, DT, LT, keys, false);
#undef KT_IX_ARGS
}

}  // namespace kt
