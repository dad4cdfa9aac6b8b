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

// ---------------------------------------------------------------------------------------------------
// kt_check_bitmap — same contract as kt_check_indexed, for selector programs whose bitmap form
// (kt_index.h) fits in LDS (the small-T regime: up to a few thousand terms).
//   phase 1, lane = pod:  8 branch-free bucket probes give the bitmap rows of the pod's labels; for each
//            word its namespace can touch (~6):  x = (rows[0] | OR_l rows[r_l])[w] & nsrows[ns][w];
//            every surviving bit is a candidate term: one TermRec read decides it.  Per-lane state is
//            ~40 registers and there is no hash-chain / posting walk.
//   phase 2, lane = (match, dimension): request rows come from the LDS tile the pods parked in phase 1.
// ---------------------------------------------------------------------------------------------------

// Everything the kernel needs, and nothing else: a compact argument block keeps the scalar register file
// free of the (large) generic table descriptors, which are reached through `sp` only on rare paths.
constexpr uint32_t kColSlots = 4;   // private match slots per pod (LDS column, plain stores)
constexpr uint32_t kWaveOvf = 64;   // shared overflow entries per wave for pods with more matches
struct BmCheckArgs {
  const uint32_t* ns;     // pod planes
  const uint32_t* flags;
  const int64_t* req;
  const uint32_t* lpair;
  const uint32_t* lkey;
  int64_t cap;
  int64_t n;
  const int64_t* rows;
  const void* recs;
  uint64_t* summary;
  uint8_t* status;
  const SelProgram* sp;   // device copy (rare term shapes, slow throttles)
  const uint8_t* ns_valid;
  const uint32_t* slow_thr;
  const void* src[6];     // LDS staging sources: rows, nsrows, nswords_off, nswords, buckets, trec
  uint32_t bytes[6];
  uint32_t off[6];        // ... and their byte offsets in LDS
  uint32_t off_cnt, off_col, off_ovq, off_req;
  uint32_t stride, bucket_mask, n_slow;
  int32_t D, L, T, dbg;
};

static BmCheckArgs make_bm_check_args(const PodTable& pods, int64_t n, const int64_t* rows, const SelProgram& sp,
                                      const SelProgram* sp_dev, const IndexDev& ix, const void* recs,
                                      uint64_t* summary, uint8_t* status, int dbg, uint32_t* total) {
  BmCheckArgs a{};
  a.ns = pods.ns, a.flags = pods.flags, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.cap = pods.cap, a.n = n, a.rows = rows, a.recs = recs, a.summary = summary, a.status = status;
  a.sp = sp_dev, a.ns_valid = sp.ns_valid, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow;
  a.D = pods.D, a.L = pods.L, a.T = sp.T, a.dbg = dbg;
  a.stride = ix.bm_stride, a.bucket_mask = ix.bm_bucket_mask;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_cnt = take(kBlockIx * 8);
  a.off_col = take(kBlockIx * kColSlots * 4);
  a.off_ovq = take((kBlockIx / kWave) * (kWaveOvf + 4) * 4);
  a.off_req = take(kBlockIx * pods.D * 8);
  const void* src[6] = {ix.bm_row_bits, ix.bm_nsrows, ix.bm_nswords_off, ix.bm_nswords, ix.bm_buckets, ix.bm_trec};
  const uint32_t bytes[6] = {ix.bm_rows * ix.bm_stride * 4, ix.bm_n_ns * ix.bm_stride * 4, (ix.bm_n_ns + 1) * 4,
                             ix.bm_n_nswords * 4, (ix.bm_bucket_mask + 1) * 32, ix.bm_n_trec * 16};
  for (int k = 0; k < 6; ++k) a.src[k] = src[k], a.bytes[k] = bytes[k], a.off[k] = take(bytes[k]);
  *total = o;
  return a;
}

// ---------------------------------------------------------------------------------------------------
// kt_check_bitmap — same contract as kt_check_indexed, for selector programs whose bitmap form
// (kt_index.h) fits in LDS (the small-T regime: up to a few thousand terms).
// WAVE-AUTONOMOUS: after the one-time staging of the tables, every wave walks its own 64-pod tiles and
// never meets a workgroup barrier again — all of a tile's state (class counters, match column, request
// rows) belongs to the wave that owns the 64 pods.
//   phase 1, lane = pod:  8 branch-free bucket probes give the bitmap rows of the pod's labels; for each
//            word its namespace can touch (~6):  x = (rows[0] | OR_l rows[r_l])[w] & nsrows[ns][w];
//            every surviving bit is a candidate term: one TermRec read decides it; a match is a plain
//            store into the pod's private LDS column (no atomics, no cross-lane traffic).
//   phase 2, lane = (match, dimension): request rows come from the wave's LDS tile.
// ---------------------------------------------------------------------------------------------------
template <int DT, int LT, bool KEYS>
__global__ __launch_bounds__(kBlockIx) void kt_check_bitmap(const BmCheckArgs a) {
  const CheckRec<DT>* recs = (const CheckRec<DT>*)a.recs;
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  lds_u32p l_rows = (lds_u32p)(lds + a.off[0]);
  lds_u32p l_nsrows = (lds_u32p)(lds + a.off[1]);
  lds_u32p l_nsw_off = (lds_u32p)(lds + a.off[2]);
  lds_u32p l_nsw = (lds_u32p)(lds + a.off[3]);
  lds_u4p l_buckets = (lds_u4p)(lds + a.off[4]);
  lds_u4p l_trec = (lds_u4p)(lds + a.off[5]);
#pragma unroll
  for (int k = 0; k < 6; ++k) lds_stage(lds + a.off[k], a.src[k], a.bytes[k]);
  __syncthreads();  // the only workgroup barrier
  const int64_t n = a.n;
  const int64_t* rows = a.rows;
  uint64_t* summary = a.summary;
  uint8_t* status = a.status;
  const int dbg = a.dbg;
  const int D = a.D, L = a.L, T = a.T;
  const int64_t cap = a.cap;
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const uint32_t stride = a.stride;
  // this wave's private LDS areas
  lds_u64wp cnt = (lds_u64wp)(lds + a.off_cnt) + wave * kWave;                 // [64] class counters
  lds_u32wp col = (lds_u32wp)(lds + a.off_col) + wave * kWave * kColSlots;     // [kColSlots][64] matches
  lds_u32wp ovq = (lds_u32wp)(lds + a.off_ovq) + wave * (kWaveOvf + 4);        // [0] = length, then entries
  KT_LDS int64_t* l_req = (KT_LDS int64_t*)(lds + a.off_req) + (size_t)wave * kWave * D;  // [64][D]
  const int64_t n_wtiles = (n + kWave - 1) / kWave;
  const int64_t wstep = (int64_t)gridDim.x * (kBlockIx / kWave);
  for (int64_t wt = (int64_t)blockIdx.x * (kBlockIx / kWave) + wave; wt < n_wtiles; wt += wstep) {
    const int64_t i = wt * kWave + lane;
    // ---- phase 1: lane = pod
    const bool in = i < n;
    const int64_t p = in ? (rows ? rows[i] : i) : 0;
    // every load of the pod's record is issued up front (one HBM round trip per tile)
    const uint32_t fl = in ? a.flags[p] : 0u;
    const uint32_t ns = in ? a.ns[p] : 0u;
    uint32_t lp[LT], lk[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      lp[l] = (in && l < L) ? a.lpair[(int64_t)l * cap + p] : 0u;
      lk[l] = (KEYS && in && l < L) ? a.lkey[(int64_t)l * cap + p] : 0u;
    }
    {  // park this pod's request row in LDS (coalesced 8*D bytes per lane) for phase 2
      int64_t myreq[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) myreq[d] = (in && d < D) ? a.req[(int64_t)p * D + d] : 0;
#pragma unroll
      for (int d = 0; d < DT; ++d)
        if (d < D) l_req[lane * D + d] = myreq[d];
    }
    cnt[lane] = 0ull;
    if (lane == 0) ovq[0] = 0u;
    const bool on = (fl & kPodValid) != 0;
    bool pod_err = false;
    uint32_t n_m = 0;
    if (on) {
      // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
      pod_err = !a.ns_valid[ns];
      const SelProgram& sp = *a.sp;
      const Matcher<LT, KEYS> m{sp, sp.ns_term_ok + (size_t)ns * sp.gw, lp, lk};
      auto classify_now = [&](uint32_t t) {  // overflow queue full (pathological match counts)
        int64_t v[DT];
        uint32_t nz = 0;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          v[d] = d < D ? a.req[(int64_t)p * D + d] : 0;
          nz |= (v[d] != 0 ? 1u : 0u) << d;
        }
        const uint32_t st = classify<DT>(recs + t, v, nz);
        if (st != 1u) lds_add64(cnt + lane, st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
        if (status) status[i * T + t] = (uint8_t)st;
      };
      auto emit = [&](uint32_t t) {
        if (dbg == 2) return;
        if (n_m < kColSlots) {
          col[n_m * kWave + lane] = t;  // private slot: fire and forget
        } else {                        // beyond the private slots: this wave's shared overflow queue
          const uint32_t pos = lds_add(ovq, 1u);
          if (pos < kWaveOvf) ovq[4 + pos] = lane << 20 | t;
          else classify_now(t);
        }
        ++n_m;
      };
      uint32_t rp[LT], rk[LT];  // word offsets of the label rows
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        rp[l] = atom_row(l_buckets, a.bucket_mask, lp[l]) * stride;
        rk[l] = KEYS ? atom_row(l_buckets, a.bucket_mask, lk[l] ? (kKeyAtom | lk[l]) : 0u) * stride : stride;
      }
      const uint32_t k1 = l_nsw_off[ns + 1];
      for (uint32_t k = l_nsw_off[ns]; k < k1; ++k) {
        const uint32_t w = l_nsw[k];
        uint32_t x = l_rows[w];  // row 0: terms without a positive requirement
#pragma unroll
        for (int l = 0; l < LT; ++l) {
          x |= l_rows[rp[l] + w];
          if (KEYS) x |= l_rows[rk[l] + w];
        }
        x &= l_nsrows[ns * stride + w];
        while (x) {
          const uint32_t c = w * 32u + (uint32_t)__ffs((int)x) - 1u;
          x &= x - 1u;
          const u32x4 tr = l_trec[c];  // {g, t, pair2, flags}
          bool ok = true;
          if (tr.w & kPostPair2) {
            bool has = false;
#pragma unroll
            for (int l = 0; l < LT; ++l) has |= lp[l] == tr.z;
            ok = has;
          }
          if (ok && (tr.w & (kPostComplex | kPostMulti))) ok = m.rare(tr.x, tr.y, tr.w);
          if (ok) emit(tr.y);
        }
      }
      // throttles with an unconvertible podSelector term: in-order walk (error semantics depend on term order)
      for (uint32_t k = 0; k < a.n_slow; ++k) {
        bool matched, err;
        const int t = (int)a.slow_thr[k];
        walk_slow<LT, KEYS>(sp, t, m.ns_row, true, lp, lk, matched, err);
        pod_err |= err;
        if (matched) emit((uint32_t)t);
      }
    }
    // ---- compaction (convergent code: bases live in scalar registers, no atomics): the parked matches of
    // row k are gathered, in place, behind those of rows < k as (pod << 20 | throttle) entries
    uint32_t n_dense = 0;
#pragma unroll
    for (uint32_t k = 0; k < kColSlots; ++k) {
      const bool has = k < n_m;
      const uint64_t mask = __ballot(has);
      const uint32_t t = has ? col[k * kWave + lane] : 0u;
      const uint32_t pos = n_dense + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      if (has) col[pos] = lane << 20 | t;
      n_dense += (uint32_t)__popcll(mask);
    }
    // ---- phase 2: lane = (match, dimension pair) over the dense entries, then the overflow entries.
    // The pod's request row (LDS) and the throttle's thr[] / head[] rows are read as 16-byte pieces.
    if (dbg != 1) {
      constexpr int LPM = DT / 2;        // lanes per match, two dimensions each
      constexpr int MPW = kWave / LPM;   // matches per iteration
      const uint32_t dp = lane % LPM, ml = lane / LPM;
      const uint64_t gmask = ((1ull << LPM) - 1ull) << (ml * LPM);
      const uint32_t n_ovf = min(ovq[0], kWaveOvf);
      const uint32_t n_items = n_dense + n_ovf;
      constexpr int U = 2;
      for (uint32_t base = 0; base < n_items; base += U * MPW) {
        uint32_t vv[U];  // (a bool array ends up in scratch memory)
        uint32_t pl[U], tt[U];
        u32x2 fa[U];
        longlong2 xx[U], th[U], hd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t j = base + u * MPW + ml;
          vv[u] = j < n_items ? 1u : 0u;
          const uint32_t e = !vv[u] ? 0u : j < n_dense ? col[j] : ovq[4 + j - n_dense];
          pl[u] = e >> 20;
          tt[u] = e & 0xFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const CheckRec<DT>* rc = recs + tt[u];
          th[u] = *(const longlong2*)(rc->thr + 2 * dp);
          hd[u] = *(const longlong2*)(rc->head + 2 * dp);
          fa[u] = *(const u32x2*)&rc->flags;  // {flags, active_mask}
          const KT_LDS int64_t* rr = l_req + pl[u] * D + 2 * dp;
          xx[u].x = (vv[u] && (int)(2 * dp) < D) ? rr[0] : 0;
          xx[u].y = (vv[u] && (int)(2 * dp + 1) < D) ? rr[1] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool nz0 = vv[u] && xx[u].x != 0, nz1 = vv[u] && xx[u].y != 0;
          const uint32_t am = fa[u].y >> (2 * dp);
          const uint64_t be = __ballot((nz0 && xx[u].x > th[u].x) || (nz1 && xx[u].y > th[u].y));
          const uint64_t bi = __ballot((nz0 && xx[u].x > hd[u].x) || (nz1 && xx[u].y > hd[u].y));
          const uint64_t ba = __ballot((nz0 && (am & 1u)) || (nz1 && (am & 2u)));
          if (dp == 0 && vv[u]) {
            const uint32_t ff = fa[u].x;
            const bool exc = (ff & kRecExceedsByCount) || (be & gmask);
            const bool act = (ff & kRecActiveByCount) || (ba & gmask);
            const bool ins = (ff & kRecInsufficientByCount) || (bi & gmask);
            const uint32_t st = exc ? 4u : act ? 2u : ins ? 3u : 1u;
            if (st != 1u) lds_add64(cnt + pl[u], st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
            if (status) status[(wt * kWave + pl[u]) * T + tt[u]] = (uint8_t)st;
          }
        }
      }
    }
    // ---- phase 3: lane = pod
    if (in) {
      const unsigned long long c = cnt[lane];
      summary[i] = !on ? 0ull : pod_err ? 2ull : (c | (c ? 1ull : 0ull));
      if (status && pod_err)
        for (int t = 0; t < T; ++t) status[i * T + t] = 255;
    }
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
    const BmCheckArgs bm_args = make_bm_check_args(pods, n, rows_dev, sp, sp_dev, ix, recs, summary, status, dbg, &bm_total);
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
