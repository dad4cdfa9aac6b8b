// kt_kernels_aggregate.hip — kt_aggregate_bitmap + kt_reduce_bitmap_slabs: per-throttle `used` through the exact term bitmaps.
#include <cstdio>
#include <type_traits>

#include "kt_scan.h"

namespace kt {

// LDS table of one chunk: one record per throttle (agg_rec_bytes, kt_index.h) — the slab a workgroup spills is the
// same bytes, so that the reduction streams whole records as 16-byte pieces
__host__ __device__ inline uint32_t agg_tab_bytes(uint32_t n_thr, int D, bool counts) { return n_thr * agg_rec_bytes(D, counts); }

// what a tile needs of its 64 pods before it can start (kt_aggregate_bitmap's fetch_tile)
// (packed fold: DT does not size a request row there — it picks the number of packed words the lane holds: 4 at DT = 8, 8 at DT = 16)
template <int DT, bool PK>
constexpr int pk_words() { return PK ? (DT > 8 ? 8 : 4) : 1; }
template <int DT, int LA, bool PK>
struct TileRecAgg {
  uint32_t p;
  uint64_t meta;
  u32x4 raw[LA / 8];
  int64_t v[PK ? 1 : DT];                      // plain fold: the request row
  unsigned long long pw[pk_words<DT, PK>()];  // packed fold: the packed words
};

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
  uint32_t off_next;  // scan views: the workgroup's next tile (its waves take tiles as they get free)
  uint32_t off_seg;  // packed fold: per word {seg_lo, seg_hi}, the run masks of throttles with several terms (built per chunk)
  uint32_t n_slow;
  int32_t D, DS, LS, T;
  int32_t counts;  // table keeps per-key pod counts instead of the presence mask
  int32_t sign;    // +1 / -1: the scanned pods are added to / removed from the target (delta scans)
  int32_t nonneg;  // no pod of the engine carries a negative request: a non-zero value then implies a non-zero sum
  int32_t has_overflow;  // some pod is flagged kMetaOverflow
  const uint64_t* v_meta;  // namespace order (ix.by_ns): scan-ordered copies, record j belongs to pod rows[j]
  const uint16_t* v_latom;
  const int64_t* v_req;
  const uint32_t* wg_range;  // nullable: record range of every workgroup, cut at namespace boundaries (plan_wg_ranges, host side)
  uint32_t* slab_tag;    // [chunks][kSlabTagStride]: epoch of the launch that last spilled the (chunk, workgroup) slab
  uint32_t epoch;
  int32_t limb;          // wide sums: which limb of every request this scan adds (limb_of, kt_device.h); 0 = the request
  uint32_t win_recs;     // 0: the table holds every record of a chunk; else the RANK WINDOW — records the LDS table holds at a time
  PackPlan pk;           // PK instantiations: the packed fold (kt_index.h)
  const uint64_t* v_pk;  //   [n_rows][pk.stride] packed request words in scan order
};

static BmAggArgs make_bm_agg_args(const PodTable& pods, const AggScan& sc, const SelProgram& sp, const SelProgram* sp_dev,
                                  const IndexDev& ix, unsigned long long* partial, unsigned char* slab, uint32_t* total) {
  BmAggArgs a{};
  a.rows = sc.rows, a.row0 = sc.row0, a.n_rows = sc.n, a.counts = sc.counts ? 1 : 0, a.sign = sc.sign, a.nonneg = sc.nonneg ? 1 : 0, a.has_overflow = sc.overflow_pods ? 1 : 0;
  a.meta = pods.meta, a.latom = pods.latom, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.sp = sp_dev, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow, a.partial = partial, a.slab = slab;
  a.D = pods.D, a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  a.slab_tag = sc.slab_tag, a.epoch = sc.epoch, a.limb = sc.limb;
  a.v_meta = sc.v_meta, a.v_latom = sc.v_latom, a.v_req = sc.v_req;
  a.wg_range = sc.by_ns && sc.wg_range_G == aggregate_blocks(sc.n) ? sc.wg_range : nullptr;
  const bool packed = sc.pk && sc.pk->nw && sc.v_pk;
  if (packed) a.pk = *sc.pk, a.v_pk = sc.v_pk;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_rank = take(ix.bm_max_words * 64u * 2u);
  // The table of the largest chunk next to the largest image — or, where that does not fit the workgroup's LDS, a RANK WINDOW
  // (round 6): the kernel scans the chunk once per window of `win_recs` ranks and folds only the matches whose record the
  // table holds at that time.  cut_chunks keeps a program in ONE chunk when only the aggregate's table stands against it (a
  // 16-dimension engine's 144-byte plain records at 1k throttles): the check then keeps its single-chunk form, the reconcile
  // pays a second scan instead of a second chunk.
  const uint32_t rec_bytes = packed ? a.pk.rec_bytes : agg_rec_bytes(pods.D, sc.counts);
  const uint32_t other = o + ((packed ? ix.bm_max_words * 16u : 0u) + 15u & ~15u) + 16u + ((ix.bm_max_lds + 15u) & ~15u);
  const uint32_t full_tab = (ix.bm_max_thr * rec_bytes + 15u) & ~15u;
  a.win_recs = 0u;
  uint32_t tab_bytes = full_tab;
  if (sc.small_window && ix.bm_max_thr > 64u) {  // (KT_AGG_SMALL_WINDOW: the parity tests run small clusters through many windows)
    a.win_recs = 64u;
    tab_bytes = (a.win_recs * rec_bytes + 15u) & ~15u;
  } else if (other + full_tab > (uint32_t)kMaxLds && other < (uint32_t)kMaxLds) {
    a.win_recs = (((uint32_t)kMaxLds - other) / rec_bytes) & ~1u;  // (even: a window's byte offset in the slab stays 16-byte aligned)
    tab_bytes = (a.win_recs * rec_bytes + 15u) & ~15u;
    if (a.win_recs < 64u) a.win_recs = 0u, tab_bytes = full_tab;  // no room for a useful window: the launch is refused below
  }
  a.off_tab = take(tab_bytes);
  a.off_seg = take(packed ? ix.bm_max_words * 16u : 0u);
  a.off_next = take(16);
  plan_bitmap_index(ix, a.ix, take);
  a.ix.by_ns = (sc.by_ns && sc.rows && sc.v_meta && sc.v_latom && (sc.v_req || packed)) ? 1u : 0u;
  *total = o;
  return a;
}

uint32_t aggregate_fixed_lds() { return 64; }

// The throttles without a rank (unconvertible selectors: the slow list) and — for a pod whose relevant atoms did not fit
// its atom row (kMetaOverflow) — EVERY throttle: walked term by term over the raw labels, straight to the partial rows
// with atomics.  Out of line on purpose: inlined, the compiler hoisted the walk's address arithmetic (label rows, term
// tables) into every tile's prologue and kept its pointers in scalar registers the hot loop then had to spill.
__device__ __attribute__((noinline)) void agg_walk_without_rank(const SelProgram* sp_, const uint32_t* slow_thr, uint32_t n_slow, int T,
                                                                const uint32_t* lpair, const uint32_t* lkey, int LS, const int64_t* req,
                                                                int D, int DS, unsigned long long* partial, int sign, int limb, uint32_t p,
                                                                uint32_t ns, bool countable, bool counted, bool overflow, uint32_t present) {
  const SelProgram& sp = *sp_;
  const int pstride = partial_stride(D);
  const uint32_t* lp = lpair + (uint64_t)p * (uint32_t)LS;
  const uint32_t* lk = lkey + (uint64_t)p * (uint32_t)LS;
  auto walk_one = [&](uint32_t t, bool lane_on) {
    const uint32_t res = walk_slow_mem(sp, (int)t, sp.ns_term_ok + (size_t)ns * sp.gw, lane_on, lp, lk, LS);
    unsigned long long* pr = partial + (size_t)t * pstride;
    if (res & kSlowError) atomicAdd(pr + partial_off_errors(D), (unsigned long long)(long long)sign);
    if ((res & kSlowMatched) && counted) {
      for (int d = 0; d < D; ++d)
        if ((present >> d) & 1u) {
          const int64_t vd = limb_of(req[(uint64_t)p * (uint32_t)DS + d], limb);
          if (vd != 0) atomicAdd(pr + d, (unsigned long long)(sign * vd));
          atomicAdd(pr + partial_off_presence(D) + d, (unsigned long long)(long long)sign);
        }
      atomicAdd(pr + partial_off_pods(D), (unsigned long long)(long long)sign);
    }
  };
  for (uint32_t ks = 0; ks < n_slow; ++ks) walk_one(slow_thr[ks], countable && !overflow);
  if (__ballot(overflow) != 0ull)
    for (int t = 0; t < T; ++t) walk_one((uint32_t)t, overflow);
}

// kt_aggregate_bitmap — `used` partials of this GPU's pod rows: affectedPods + fold Add for all throttles
// (throttle_controller.go:116-119,221-246; clusterthrottle_controller.go:119-122,224-270).
// The workgroup walks the chunks of the index: chunk image in, table of the chunk's throttles zeroed, every tile of
// the workgroup scanned against it — wave-autonomous like kt_check_bitmap, lane = pod throughout: the lane holds its
// pod's request row in registers (loaded for counted pods only) and folds ResourceAmountOfPod into the throttle's record
// of the LDS table (v i64[D] | presence mask u32 | pods u32) for every term scan_tile reports (ds_add_u64 per non-zero dimension) — then the table is spilled to this
// (chunk, workgroup)'s slab; kt_reduce_bitmap_slabs sums the slabs into the partial buffer.
// No global atomics except for throttles with unconvertible selectors (the "slow" list).
// PK: the packed fold — the lane holds its pod's contribution as 1..4 packed words (PackPlan; 5..8 in the DT = 16 instantiations:
//     engines with more than 8 dimensions) and adds whole words: one
//     ds_add_u64 where the plain fold issues one per non-zero dimension plus the pod count; the record is nw words + the
//     OR of the key masks of pods that carry a key with the value 0.  Full scans over the scan view only (no counts mode,
//     no negative requests, sign +1).
// (One workgroup per CU: two packed ones fit when the record is squeezed to 32 bytes, but the 64-VGPR form spills and the
//  32-byte stride lands 8 records on one bank group — measured 33 us against 24 us at 1M x 1k.)
// WIN: the table holds a WINDOW of the chunk's records at a time (BmAggArgs::win_recs) and the tiles are scanned once per window;
//      its own instantiation, so that the kernels whose tables fit (every BASELINE configuration) do not pay the window test of
//      every fold step — measured as one loop with a run-time window: aggregate 0.458 -> 0.481 ms on the configs[4] shard.
template <int DT, int LA, bool VETO, int NEED, bool PK, bool WIN = false>
__global__ __launch_bounds__(kBlockIx) void kt_aggregate_bitmap(const BmAggArgs a) {
  const int D = a.D, DS = a.DS;
  // QUEUED FOLD (round 5, the packed instantiations).  Folding a word's matches where the scan produced them steps as often as
  // the lane with the MOST matches of that word has matches — lanes 21 % (configs[2]) to 38 % (configs[4]) busy — and every
  // step pays the bit extraction (~16 VALU instructions), a rank read and three LDS atomics.  Measured on the configs[4] shard
  // (timing probes, profiles/r05_probe_breakdown.txt): of the aggregate's 0.70 ms the scan is 0.30, the extraction loop 0.18,
  // the atomics 0.22.  The lane therefore QUEUES what the scan finds and the wave folds only when some lane's queue is full
  // or the tile is done — a step of the fold then serves nearly every lane that has anything left:
  // the match words themselves, (word number, 64 bits) x kWq — one predicated push per visit, the extraction runs inside the
  // balanced steps: 0.53 -> 0.48 ms on the shard.  (A queue of 16-bit term numbers halved the atomic instructions as well but kept
  // the lopsided extraction loop: 0.53 ms; it and the other A/B forms of round 5 are in the git history of this file.)
  constexpr bool kFoldQueue = PK && LA <= 8;  // (16 / 32 atom slots: the queue does not fit the registers — 40 B of scratch at 16)
  constexpr int NW = pk_words<DT, PK>();
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: LDS bases stay scalar
  // pod rows and list positions fit 32 bits (pod_capacity <= 2^31): no 64-bit index arithmetic in the tile loop
  const uint32_t n_rows = (uint32_t)a.n_rows;
  const uint32_t n_wtiles = (uint32_t)(((uint64_t)n_rows + kWave - 1) / kWave);
  const uint32_t wstep = gridDim.x * (uint32_t)(kBlockIx / kWave);
  KT_LDS const uint16_t* trank = (KT_LDS const uint16_t*)(lds + a.off_rank);
  const bool counts = a.counts != 0;
  // namespace order (a.ix.by_ns): this workgroup owns the tiles [t_lo, t_hi) and only walks — and only spills the
  // tables of — the chunks that hold words of their namespaces
  // (the packed fold only runs over the namespace-ordered scan view: known when the kernel is compiled — every record is
  //  then read at its list position, nothing hangs off the row list any more)
  const bool by_ns = PK || a.ix.by_ns != 0u;
  uint32_t t_lo = 0, t_hi = n_wtiles;
  uint32_t ns_lo = 0, ns_hi = 0;
  // the records the workgroup's tiles are cut from: [rec0, rec_end) — everything, or this workgroup's planned range of a
  // namespace-ordered view (ends at a namespace boundary where one lies close: see plan_wg_ranges, kt_kernels.hip)
  uint32_t rec0 = 0, rec_end = n_rows;
  if (by_ns && a.wg_range) {
    rec0 = __builtin_amdgcn_readfirstlane(a.wg_range[blockIdx.x]), rec_end = __builtin_amdgcn_readfirstlane(a.wg_range[blockIdx.x + 1u]);
    if (rec0 >= rec_end) return;  // (multi-chunk programs only: their reductions skip a slab nobody tagged)
    t_lo = 0u, t_hi = (rec_end - rec0 + kWave - 1u) / kWave;
    ns_lo = (uint32_t)(a.v_meta[rec0] & kMetaNsMask);
    ns_hi = (uint32_t)(a.v_meta[rec_end - 1u] & kMetaNsMask);
    ns_lo = __builtin_amdgcn_readfirstlane(ns_lo), ns_hi = __builtin_amdgcn_readfirstlane(max(ns_hi, ns_lo));
  } else if (by_ns) {
    const uint32_t tpb = (n_wtiles + gridDim.x - 1u) / gridDim.x;
    t_lo = min(blockIdx.x * tpb, n_wtiles), t_hi = min(t_lo + tpb, n_wtiles);
    // A workgroup without tiles must NOT return: the reductions of single-chunk programs read every launched
    // workgroup's slab without looking at tags, so it spills its zeroed table like everybody else (round 3 returned
    // here and left the slab as the allocator or an earlier, larger scan had it).  launch_aggregate_indexed sizes the
    // grid so that no workgroup is empty; this is the second line of defence.
    if (t_lo < t_hi) {
      ns_lo = (uint32_t)(a.v_meta[(uint64_t)t_lo * kWave] & kMetaNsMask);
      ns_hi = (uint32_t)(a.v_meta[min((uint64_t)t_hi * kWave, (uint64_t)n_rows) - 1u] & kMetaNsMask);
    }
    ns_lo = __builtin_amdgcn_readfirstlane(ns_lo), ns_hi = __builtin_amdgcn_readfirstlane(max(ns_hi, ns_lo));
  }
  // (the walk without a rank — slow list, overflow pods — rides on the first chunk the workgroup stages)
  uint32_t first_ci = 0u, last_ci = a.ix.n_chunks - 1u;
  if (by_ns) relevant_chunk_span(a.ix, ns_lo, ns_hi, first_ci, last_ci), last_ci = max(last_ci, first_ci);
  for (uint32_t ci = first_ci; ci <= last_ci; ++ci) {
    if (by_ns && ci != first_ci && !chunk_relevant(a.ix, ci, ns_lo, ns_hi)) continue;
    const BmChunk ch = a.ix.chunks[ci];
    const uint32_t n_thr = ch.n_thr;
    const uint32_t rec = PK ? a.pk.rec_bytes : agg_rec_bytes(D, counts);
    const uint32_t slab_pitch = (n_thr * rec + 15u) & ~15u;  // a workgroup's records of this chunk in the slab scratch
    KT_LDS unsigned char* tab = lds + a.off_tab;
    // the rank window (make_bm_agg_args): all records at once, or win_recs of them per pass over the workgroup's tiles
    const uint32_t win = WIN ? a.win_recs : (n_thr ? n_thr : 1u);
    for (uint32_t r0 = 0; r0 == 0u || (WIN && r0 < n_thr); r0 += win) {
      const uint32_t nrec = WIN ? (n_thr > r0 ? min(win, n_thr - r0) : 0u) : n_thr;
      const uint32_t tab_bytes = (nrec * rec + 15u) & ~15u;
      // the tile's records — meta word, atom row AND the request words: every lane's, so that the request does not hang off
      // the meta word by another trip to memory
      const uint32_t wt0 = by_ns ? t_lo + wave : blockIdx.x * (uint32_t)(kBlockIx / kWave) + wave;
      const uint32_t wt_step = by_ns ? (uint32_t)(kBlockIx / kWave) : wstep;
      auto fetch_tile = [&](uint32_t wt) {
        TileRecAgg<DT, LA, PK> r;
        const uint32_t ic = min(rec0 + wt * kWave + lane, rec_end - 1u);
        r.p = a.rows ? (uint32_t)a.rows[ic] : (uint32_t)a.row0 + ic;
        r.meta = by_ns ? a.v_meta[ic] : a.meta[r.p];
        load_atoms<LA>(by_ns ? a.v_latom : a.latom, by_ns ? ic : r.p, r.raw);
        if constexpr (!PK) {
          load_requests<DT>(by_ns ? a.v_req : a.req, DS, by_ns ? ic : r.p, r.v);
        } else {
          const u64x2* q = (const u64x2*)(a.v_pk + (uint64_t)ic * a.pk.stride);
          const u64x2 q0 = q[0];
          r.pw[0] = q0.x, r.pw[1] = q0.y, r.pw[2] = 0ull, r.pw[3] = 0ull;
          if constexpr (NW > 4) {  // (these instantiations only run plans of 5..8 words: stride 8)
            const u64x2 q1 = q[1], q2 = q[2], q3 = q[3];
            r.pw[2] = q1.x, r.pw[3] = q1.y, r.pw[4] = q2.x, r.pw[5] = q2.y, r.pw[6] = q3.x, r.pw[7] = q3.y;
          } else if (a.pk.stride > 2u) {
            const u64x2 q1 = q[1];
            r.pw[2] = q1.x, r.pw[3] = q1.y;
          }
        }
        return r;
      };
      TileRecAgg<DT, LA, PK> cur{};
      __syncthreads();  // nobody reads the previous image / table any more
      for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
      if (by_ns && threadIdx.x == 0) *(lds_u32wp)(lds + a.off_next) = t_lo;
      if (r0 == 0u) {  // the image and the ranks of the chunk's term numbers: one batch of loads (they stay for the later windows)
        const StageSeg segs[2] = {chunk_image_segment(a.ix, ch),
                                  StageSeg{a.off_rank, (const u32x4*)(a.ix.blob + ch.img_off + ch.off_term_rank), ch.n_words * 8u}};
        lds_stage_segments<2>(lds, segs);
      }
      if constexpr (kFoldQueue) {
        // the run masks of the chunk's words, by ballot from the rank words (a wave per word, lane = term number): lowest /
        // highest number of every run of one group's terms — a throttle with several terms is counted once, and with the
        // masks that rule is applied to a whole word at a time (as kt_check_bitmap's WordVerdict does), so that the order
        // in which a lane's matches are folded no longer matters
        if (ch.has_adj && r0 == 0u) {
          const uint16_t* g_rank = (const uint16_t*)(a.ix.blob + ch.img_off + ch.off_term_rank);
          for (uint32_t w = wave; w < ch.n_words; w += (uint32_t)(kBlockIx / kWave)) {
            const uint32_t tr = g_rank[w * 64u + lane];
            const uint32_t tp = (uint32_t)__shfl_up((int)tr, 1), tn = (uint32_t)__shfl_down((int)tr, 1);
            const bool adj = (tr & kRankAdj) != 0u;
            const bool same_prev = lane > 0u && adj && tp == tr, same_next = lane < 63u && adj && tn == tr;
            const uint64_t m_lo = __ballot(!same_prev), m_hi = __ballot(!same_next);
            if (lane < 2u) ((lds_u64wp)(lds + a.off_seg))[w * 2u + lane] = lane == 0u ? m_lo : m_hi;
          }
        }
      }
      const BmView bm = open_chunk<VETO>(lds, a.ix, ch);
      __syncthreads();
      auto next_tile = [&](uint32_t prev) -> uint32_t {  // (see kt_check_bitmap)
        if (!by_ns) return prev + wt_step;
        uint32_t t = 0u;
        if (lane == 0u) t = lds_add((lds_u32wp)(lds + a.off_next), 1u);
        return __builtin_amdgcn_readfirstlane(t);
      };
      for (uint32_t wt = by_ns ? next_tile(0u) : wt0; wt < t_hi; wt = next_tile(wt)) {
        // ---- the tile's records, always from valid addresses (lanes past the end re-read the last row and are off);
        //      requested before the chunk was staged / behind the previous tile's peel (fetch_tile)
        cur = fetch_tile(wt);
        const uint32_t i = rec0 + wt * kWave + lane;
        const bool in = i < rec_end;
        const uint32_t p = cur.p;
        const uint64_t meta = cur.meta;
        u32x4 raw[LA / 8];
#pragma unroll
        for (int q = 0; q < LA / 8; ++q) raw[q] = cur.raw[q];
        const uint32_t st = (uint32_t)(meta >> kMetaStateShift) & 0xFu;
        // shouldCountIn (throttle_controller.go:217-219); terminated pods are matched but not counted
        // (isNotFinished, pod_util.go:26-28) and only matter for error detection (slow list)
        const bool countable = in && (st & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
        const bool counted = countable && !(st & kPodFinished);
        if (__ballot(countable) == 0ull) continue;  // (wave-uniform: nobody of the tile counts)
        const uint32_t ns = countable ? (uint32_t)(meta & kMetaNsMask) : 0u;
        const uint32_t present = (uint32_t)(meta >> kMetaPresentShift) & 0xFFFFu;
        // kt_finalize calls a key present when its contributor count OR its sum is non-zero: the presence mask only has
        // to travel for keys this pod carries with the value 0 — unless negative requests exist (sums can cancel)
        const bool need_pres = !a.nonneg || (present & ~(uint32_t)(meta >> kMetaNzShift)) != 0u;
        // ResourceAmountOfPod: the request row (or its packed words) travelled with the record — every lane's, so that
        // the request does not hang off the meta word by another trip to memory.  What a pod that is not counted brought
        // is never looked at: its lane takes no part in the scan (scan_counted), so no match is ever handed to it.
        int64_t v[DT];             // plain fold (dead in the PK instantiations)
        unsigned long long pw[NW > 4 ? NW : 4];  // packed fold (dead in the others)
#pragma unroll
        for (int d = 0; d < DT; ++d) v[d] = 0;
#pragma unroll
        for (int k = 0; k < (NW > 4 ? NW : 4); ++k) pw[k] = 0ull;
        if constexpr (!PK) {
#pragma unroll
          for (int d = 0; d < DT; ++d) v[d] = limb_of(cur.v[d], a.limb);
        } else {
#pragma unroll
          for (int k = 0; k < NW; ++k) pw[k] = cur.pw[k];
        }
        const uint32_t zero_keys = present & ~(uint32_t)(meta >> kMetaNzShift) & 0xFFFFu;  // keys carried with the value 0
        uint32_t ro[LA];
        atom_row_offsets<LA>(raw, ro);

        // ---- throttles with unconvertible selectors have no rank: walked once (with the first chunk), straight to the
        //      result buffer.  So is EVERY throttle for a pod whose relevant atoms did not fit its atom row (kMetaOverflow).
        const bool overflow = a.has_overflow && countable && (meta & kMetaOverflow) != 0;
        const bool scan_counted = counted && !overflow;
        if (ci == first_ci && r0 == 0u && (a.n_slow || a.has_overflow))
          agg_walk_without_rank(a.sp, a.slow_thr, a.n_slow, a.T, a.lpair, a.lkey, a.LS, a.req, D, DS, a.partial, a.sign, a.limb, p, ns, countable,
                                counted, overflow, present);

        uint32_t last_r = 0xFFFFFFFFu;
        const uint32_t pk_nw = __builtin_amdgcn_readfirstlane(a.pk.nw);
        // the lane's packed words onto a record: every word of the plan (wave-uniform count — 5..8 in the NW = 8 forms, whose
        // first five are unconditional)
        auto add_words = [&](lds_u64wp tv) {
          if constexpr (NW > 4) {
#pragma unroll
            for (int k = 0; k < 5; ++k) lds_add64(tv + k, pw[k]);
            if (pk_nw > 5u) lds_add64(tv + 5, pw[5]);
            if (pk_nw > 6u) lds_add64(tv + 6, pw[6]);
            if (pk_nw > 7u) lds_add64(tv + 7, pw[7]);
          } else {
            lds_add64(tv, pw[0]);
            if (pk_nw > 1u) lds_add64(tv + 1, pw[1]);
            if (pk_nw > 2u) lds_add64(tv + 2, pw[2]);
            if (pk_nw > 3u) lds_add64(tv + 3, pw[3]);
          }
        };
        // one matched term number of the lane's pod, given its rank word
        auto add_match = [&](bool has, uint32_t tr) {
              const uint32_t r = tr & 0x7FFFu;  // chunk-local throttle rank
              // a throttle with several terms is counted once
              const bool ok = has && !((tr & kRankAdj) && r == last_r);
              if (ok) last_r = r;
              if (ok && (!WIN || r - r0 < nrec)) {  // (WIN: is the record in the table during this window — wave-uniform bounds, unsigned compare)
                KT_LDS unsigned char* rp = tab + __umul24(WIN ? r - r0 : r, rec);  // the throttle's record (rank < 2^15, record <= 272 bytes)
                lds_u64wp tv = (lds_u64wp)rp;
                if constexpr (PK) {
                  // every word of the plan, whatever it holds (the number of words is wave-uniform: scalar branches): an
                  // LDS atomic costs per instruction, not per lane, and some lane of the step always has a non-zero word —
                  // testing the words lane by lane only bought exec-mask juggling
                  add_words(tv);
                  if (zero_keys) (void)__hip_atomic_fetch_or((lds_u32wp)(rp + a.pk.nw * 8u), zero_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  return;
                }
                lds_u32wp tu = (lds_u32wp)(rp + (uint32_t)D * 8);
#pragma unroll
                for (int d = 0; d < DT; ++d)
                  if (v[d] != 0) lds_add64(tv + d, (unsigned long long)v[d]);  // padding dimensions hold 0
                if (counts) {
#pragma unroll
                  for (int d = 0; d < DT; ++d)
                    if ((present >> d) & 1u) lds_add(tu + d, 1u);
                  lds_add(tu + D, 1u);
                } else {
                  if (need_pres) (void)__hip_atomic_fetch_or(tu, present, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  lds_add(tu + 1, 1u);
                }
              }
        };
        auto confirm_slow = [&](uint32_t c) {
          return term_match_mem(*a.sp, bm.term_g[c], a.lpair + (uint64_t)p * (uint32_t)a.LS, a.lkey + (uint64_t)p * (uint32_t)a.LS, a.LS);
        };
        if constexpr (kFoldQueue) {
          // WORD queue: the lane keeps the match words of the last visits as they are — (word number, 64 match bits), kWq of them,
          // newest first — and nothing is extracted while the scan runs: ONE predicated push per visit.  When some lane's queue is
          // full (or the tile is done) the wave folds: every step each lane that has anything takes the lowest bit of its newest
          // word, so a step serves nearly every lane that has matches left — and the bit extraction, which the term-number queue
          // pays in a loop that runs as often as the BUSIEST lane of every single word has matches, runs in these balanced steps too.
          constexpr int kWq = 4;  // (measured on the configs[4] shard: 2 words 0.512, 3: 0.484, 4: 0.476, 5: 0.483 ms)
          uint64_t qx[kWq];
          // word numbers, 10 bits each (newest lowest: cut_chunks keeps a chunk below 1024 words); entries in use
          typedef typename std::conditional<(kWq > 3), uint64_t, uint32_t>::type qw_t;
          static_assert(kWq * 10 <= 64, "ten bits per queued word number");
          qw_t qw = 0;
          uint32_t qn = 0u;
#pragma unroll
          for (int k = 0; k < kWq; ++k) qx[k] = 0ull;
          auto add_rank = [&](bool has, uint32_t r) {
            if (has && (!WIN || r - r0 < nrec)) {
              KT_LDS unsigned char* rp = tab + __umul24(WIN ? r - r0 : r, rec);
              lds_u64wp tv = (lds_u64wp)rp;
              add_words(tv);
              if (zero_keys) (void)__hip_atomic_fetch_or((lds_u32wp)(rp + a.pk.nw * 8u), zero_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          };
          auto flush = [&]() {
            while (__ballot(qn != 0u) != 0ull) {
              const bool has = qn != 0u;
              const uint32_t cw = ((uint32_t)qw & 1023u) * 64u;
              const uint32_t c = cw + (uint32_t)__ffsll((unsigned long long)qx[0]) - 1u;
              qx[0] &= qx[0] - 1ull;
              const uint32_t r = trank[has ? c : 0u] & 0x7FFFu;
              if (has && qx[0] == 0ull) {  // the newest word is used up: the older ones move up
#pragma unroll
                for (int k = 0; k + 1 < kWq; ++k) qx[k] = qx[k + 1];
                qx[kWq - 1] = 0ull;
                qw >>= 10;
                qn -= 1u;
              }
              add_rank(has, r);
            }
          };
          const bool seg_on = ch.has_adj != 0u;  // wave-uniform: some throttle of the chunk has several terms
          KT_LDS const u64x2* segp = (KT_LDS const u64x2*)(lds + a.off_seg);
          scan_tile<LA, VETO, NEED, VETO>(
              bm, scan_counted, ns, ro, [&](bool, uint32_t) {}, confirm_slow,
              [&](uint32_t w, uint64_t x, const u64x2& seg) -> uint64_t {
                if (seg_on) {  // a throttle with several terms is counted once: the lowest match of every run
                  const uint64_t v = x | seg.y;
                  x = andn_64(x, v - seg.x);
                }
                if (__ballot(x != 0ull && qn >= (uint32_t)kWq) != 0ull) flush();
                if (x != 0ull) {
#pragma unroll
                  for (int k = kWq - 1; k > 0; --k) qx[k] = qx[k - 1];
                  qx[0] = x;
                  qw = (qw_t)(qw << 10) | (qw_t)w;
                  qn += 1u;
                }
                return 0ull;
              },
              [&](uint32_t w) -> u64x2 { return seg_on ? segp[w] : u64x2{0ull, 0ull}; });
          flush();
        } else if constexpr (PK) {
          // the packed fold takes a word's matches where the scan produced them (scan_tile's post hook), TWO per step: both
          // rank reads are in flight together and the wave steps ceil(matches / 2) times per word instead of once per match
          // through scan_tile's peel (ascending term numbers per lane, as the run rule of add_match needs)
          scan_tile<LA, VETO, NEED, VETO>(
              bm, scan_counted, ns, ro, [&](bool, uint32_t) {}, confirm_slow,
              [&](uint32_t w, uint64_t x, int) -> uint64_t {
                uint64_t xf = x;
                while (__ballot(xf != 0ull) != 0ull) {
                  const bool h1 = xf != 0ull;
                  const uint32_t c1 = h1 ? w * 64u + (uint32_t)__ffsll((unsigned long long)xf) - 1u : 0u;
                  xf &= xf - 1ull;
                  const bool h2 = xf != 0ull;
                  const uint32_t c2 = h2 ? w * 64u + (uint32_t)__ffsll((unsigned long long)xf) - 1u : 0u;
                  xf &= xf - 1ull;
                  const uint32_t tr1 = trank[c1], tr2 = trank[c2];
                  add_match(h1, tr1);
                  add_match(h2, tr2);
                }
                return 0ull;
              });
        } else {
          scan_tile<LA, VETO, NEED, VETO>(
              bm, scan_counted, ns, ro, [&](bool has, uint32_t c) { add_match(has, trank[c]); }, confirm_slow);
        }
      }
      __syncthreads();  // spill this (chunk, workgroup)'s table — this window of it — : coalesced 16-byte stores
      u32x4* dst = (u32x4*)(a.slab + (size_t)ch.slab_off * 16 + (size_t)blockIdx.x * slab_pitch + (size_t)r0 * rec);
      lds_u4p src = (lds_u4p)(lds + a.off_tab);
      for (uint32_t i = threadIdx.x; i < tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
    }  // rank windows
    if (threadIdx.x == 0) a.slab_tag[ci * kSlabTagStride + blockIdx.x] = a.epoch;  // the reduction skips slabs this launch left alone
  }
}

// kt_reduce_bitmap_slabs — partial[t][*] += sum over the workgroups' slabs of t's chunk.  Every slab is an array of
// per-throttle records (agg_rec_bytes).  grid = (piece groups, chunks, slab splits): a wave streams 64 consecutive
// 16-byte pieces of the chunk's table (1 KB, fully coalesced) over its share of the slabs, the block's four waves meet
// in LDS, and the owner of a piece adds its words to the partial buffer with atomics (kSlabSplits blocks per piece —
// the split is what puts every CU on the stream): values -> [0,D), key seen (>= 1) or per-key pod counts -> [D,2D),
// pods -> 2D.  Throttles of the slow list were written by the scan kernel with atomics too, and so was every error
// word (2D+1), which is left alone.
constexpr int kSlabSplits = 4;
__global__ __launch_bounds__(256) void kt_reduce_bitmap_slabs(const unsigned char* slab, const BmChunk* chunks,
                                                              const uint32_t* rank_t, int n_slabs, int D, int counts, int sign,
                                                              const uint32_t* slab_tag, uint32_t epoch, unsigned long long* partial) {
  constexpr int G = 4;  // waves per block
  __shared__ unsigned long long part[G][64][4];
  const BmChunk ch = chunks[blockIdx.y];
  const uint32_t rec = agg_rec_bytes(D, counts != 0), ppr = rec / 16u;  // pieces per record
  const uint32_t n_pieces = ch.n_thr * ppr;
  if (blockIdx.x * 64u >= n_pieces) return;
  const size_t pitch = (size_t)ch.n_thr * rec;
  const uint32_t wl = threadIdx.x & 63u, g = threadIdx.x >> 6;
  const uint32_t pi = blockIdx.x * 64u + wl;
  const bool in = pi < n_pieces;
  const uint32_t r = in ? pi / ppr : 0u, q = in ? pi - r * ppr : 0u;
  // kind of the piece's four dwords: 0 = half of an int64 value (the pair is added as one), 1 = u32 sum, 2 = u32 OR, 3 = padding
  const uint32_t dw0 = q * 4u, n_val = 2u * (uint32_t)D, n_u32 = counts ? (uint32_t)D + 1u : 2u;
  uint32_t kind[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t dw = dw0 + k;
    kind[k] = dw < n_val ? 0u : dw < n_val + n_u32 ? ((!counts && dw == n_val) ? 2u : 1u) : 3u;
  }
  unsigned long long acc[4] = {0, 0, 0, 0};
  const unsigned char* base = slab + (size_t)ch.slab_off * 16 + (size_t)(in ? pi : 0u) * 16;
  const int step = G * kSlabSplits;
  const uint32_t* tag = slab_tag + blockIdx.y * kSlabTagStride;
#pragma unroll 8
  for (int b = (int)(blockIdx.z * G + g); b < n_slabs; b += step) {
    if (tag[b] != epoch) continue;  // that workgroup had no pods for this chunk (namespace-ordered scans)
    const u32x4 x = *(const u32x4*)(base + (size_t)b * pitch);
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (kind[2 * h] == 0u) {
        acc[2 * h] += (unsigned long long)w[2 * h] | (unsigned long long)w[2 * h + 1] << 32;
      } else {
        acc[2 * h] = kind[2 * h] == 2u ? (acc[2 * h] | w[2 * h]) : acc[2 * h] + w[2 * h];
        acc[2 * h + 1] = kind[2 * h + 1] == 2u ? (acc[2 * h + 1] | w[2 * h + 1]) : acc[2 * h + 1] + w[2 * h + 1];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[g][wl][k] = acc[k];
  __syncthreads();
  if (g == 0 && in) {
    const int stride = partial_stride(D);
    unsigned long long* prow = partial + (size_t)rank_t[ch.rank0 + r] * stride;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (kind[k] == 3u || (kind[k] == 0u && (k & 1))) continue;
      unsigned long long sum = 0;
#pragma unroll
      for (int gg = 0; gg < G; ++gg) sum = kind[k] == 2u ? (sum | part[gg][wl][k]) : sum + part[gg][wl][k];
      if (sum == 0ull) continue;
      const uint32_t dw = dw0 + k;
      if (kind[k] == 0u) {
        atomicAdd(prow + dw / 2u, (unsigned long long)((long long)sign * (long long)sum));
      } else if (kind[k] == 2u) {  // presence mask: key seen by some slab of this split
        for (int d = 0; d < D; ++d)
          if ((sum >> d) & 1ull) atomicAdd(prow + partial_off_presence(D) + d, 1ull);
      } else {
        const uint32_t u = dw - n_val;  // counts mode: u < D per-key pod counts, u == D pods; mask mode: u == 1 pods
        const int j = counts ? (u < (uint32_t)D ? partial_off_presence(D) + (int)u : partial_off_pods(D)) : partial_off_pods(D);
        atomicAdd(prow + j, (unsigned long long)((long long)sign * (long long)sum));
      }
    }
  }
}

// kt_reduce_packed_slabs — the same for slabs of PACKED records (PackPlan).  One block of 16 waves per tile of whole
// records of a chunk's slab row: block_record_sums (kt_index_device.h) sums the tile over the workgroups' slabs — coalesced,
// whole words, class by class — into LDS; then thread = (record, dimension) cuts its total out of the sums and adds it to
// the partial buffer (several groups of one throttle meet there): a dozen atomics per record, issued by a dozen lanes at
// once.  check_tags = 0: every workgroup spilled every chunk (single-chunk programs) — no tag reads.
__global__ __launch_bounds__(kRecBlock) void kt_reduce_packed_slabs(const unsigned char* slab, const BmChunk* chunks, const uint32_t* rank_t,
                                                                   int n_slabs, int D, const PackPlan pk, const uint32_t* slab_tag, uint32_t epoch,
                                                                   int check_tags, unsigned long long* partial) {
  __shared__ RecSumsLds lds;
  const BmChunk ch = chunks[blockIdx.y];
  const uint32_t units = pk.rec_bytes >> 3, rb = (uint32_t)kRecTileUnits / units;
  const uint32_t rec0 = blockIdx.x * rb;
  if (rec0 >= ch.n_thr) return;  // block-uniform
  const uint32_t nrec = min(rb, ch.n_thr - rec0);
  const uint32_t x = threadIdx.x, g = x >> 4, d = x & 15u;  // thread = (record g of the tile, dimension d)
  const size_t pitch = ((size_t)ch.n_thr * pk.rec_bytes + 15u) & ~(size_t)15u;
  RecSlabLoads sl;
  record_slabs_live(n_slabs, slab_tag + blockIdx.y * kSlabTagStride, epoch, check_tags, sl);
  uint32_t t = 0;
  if (g < nrec) t = rank_t[ch.rank0 + rec0 + g];
  record_slabs_issue(slab + (size_t)ch.slab_off * 16 + (size_t)rec0 * pk.rec_bytes, pitch, nrec * units, sl);
  block_record_sums(sl, pk, lds);
  if (g >= nrec) return;
  const uint32_t ub = g * units;
  const unsigned long long pods = packed_pods(lds, ub, pk);
  if (pods == 0ull) return;  // nobody matched this throttle
  unsigned long long* prow = partial + (size_t)t * partial_stride(D);
  if (d == 0u) atomicAdd(prow + partial_off_pods(D), pods);
  if (d < (uint32_t)D) {
    const unsigned long long mine = packed_field(lds, ub, packed_desc_of(pk, (int)d, D));
    if (mine) atomicAdd(prow + d, mine);
    // key seen: a non-zero sum says so by itself (kt_finalize); a key only ever carried with the value 0 is marked here
    if ((packed_zero_keys(lds, ub, pk) >> d) & 1u) atomicAdd(prow + partial_off_presence(D) + d, 1ull);
  }
}

#define KT_AGG_BM_LAUNCH(DT_, LA_, VETO_, NEED_, PK_, WIN_)                                                    \
  {                                                                                                           \
    auto kfn = kt_aggregate_bitmap<DT_, LA_, VETO_, NEED_, PK_, WIN_>;                                        \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bm);     \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bm, s, bm_args);                                                      \
  }
#define KT_AGG_BM_CASE(DT_, LA_, VETO_, NEED_)                                                                \
  {                                                                                                           \
    if (!packed && !windowed) KT_AGG_BM_LAUNCH(DT_, LA_, VETO_, NEED_, false, false)                          \
    else if (!packed) KT_AGG_BM_LAUNCH(DT_, LA_, VETO_, NEED_, false, true)                                   \
    else if (!windowed && !wide_pk) KT_AGG_BM_LAUNCH(8, LA_, VETO_, NEED_, true, false)                      \
    else if (!wide_pk) KT_AGG_BM_LAUNCH(8, LA_, VETO_, NEED_, true, true)                                     \
    else if (!windowed) KT_AGG_BM_LAUNCH(16, LA_, VETO_, NEED_, true, false)                                  \
    else KT_AGG_BM_LAUNCH(16, LA_, VETO_, NEED_, true, true)                                                  \
  }

static_assert(kCUs <= kMaxSlabsPerRecord, "packed_record_sums takes four slabs per lane");
int aggregate_blocks(int64_t n_rows) {
  int64_t b = (n_rows + kBlockIx - 1) / kBlockIx;
  return (int)(b < 1 ? 1 : b > kCUs ? kCUs : b);
}
// the most pods one workgroup of an aggregate launch scans: contiguous tile ranges (scan view) or a stride over the tiles
uint64_t aggregate_slab_pods(int64_t n_rows, int blocks) {
  const int64_t tiles = (n_rows + kWave - 1) / kWave;
  const int64_t per_wg = (tiles + blocks - 1) / blocks + (kBlockIx / kWave);  // either assignment, rounded up generously
  return (uint64_t)per_wg * kWave;
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
  uint32_t bm_total = 0;
  BmAggArgs bm_args = make_bm_agg_args(pods, sc, sp, sp_dev, ix, partial, slab, &bm_total);
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  // the packed fold: full scans over the scan view, records no larger than the plain ones (the slab areas were sized for those)
  // (ix.cut_thr_bytes: the record size the chunks' tables and slab areas were cut for — the plain record, or the packed fold's
  //  for a program of several chunks; the engine cuts again before it launches a fold whose records are larger)
  // (ix.has_long: the packed fold applies "counted once" per word — a throttle whose run spans words takes the plain fold)
  const bool packed = bm_args.v_pk != nullptr && bm_args.ix.by_ns && !sc.counts && sc.sign == 1 && sc.nonneg &&
                      bm_args.pk.rec_bytes <= ix.cut_thr_bytes && !ix.has_long;
  if (bm_args.v_pk != nullptr && !packed) return nullptr;  // the engine only hands over packed words it may use
  if (!packed && agg_rec_bytes(pods.D, sc.counts) > ix.cut_thr_bytes) return nullptr;
  int nb = aggregate_blocks(n_rows);
  if (bm_args.ix.by_ns && bm_args.wg_range) {
    // planned ranges: one per workgroup of the full grid (a workgroup whose range is empty returns: multi-chunk programs only,
    // whose reductions go by the slab tags)
    if (ix.n_chunks == 1) bm_args.wg_range = nullptr;
  }
  if (bm_args.ix.by_ns && !bm_args.wg_range) {
    // contiguous tile ranges of ceil(tiles / nb) tiles: launch exactly the workgroups that own at least one tile (with
    // 10 163 tiles and 256 workgroups the last one would own none).  ceil(tiles / nb) is the same for the smaller grid.
    const int64_t tiles = (n_rows + kWave - 1) / kWave, tpb = (tiles + nb - 1) / nb;
    nb = (int)((tiles + tpb - 1) / tpb);
  }
  dim3 g_(nb), b_(kBlockIx);
  const size_t lds_bm = bm_total;
  const bool windowed = bm_args.win_recs != 0u;
  const bool wide_pk = packed && bm_args.pk.nw > 4u;  // 5..8 packed words: the DT = 16 instantiations of the packed fold
  static const bool dbg_lds = getenv("KT_DEBUG_LDS") != nullptr;
  if (dbg_lds)
    fprintf(stderr, "kt_aggregate_bitmap: lds=%u chunks=%u largest LDS part=%u max thr=%u T=%d packed=%d nw=%u rec=%u\n", bm_total, ix.n_chunks,
            ix.bm_max_lds, ix.bm_max_thr, sp.T, packed ? 1 : 0, bm_args.pk.nw, bm_args.pk.rec_bytes);
#ifdef KT_FAST_BUILD
  KT_AGG_BM_CASE(8, 8, false, 2)
#else
  if (!ix.rich) { if (DT <= 8) KT_AGG_BM_CASE(8, 8, false, 2) else KT_AGG_BM_CASE(16, 8, false, 2) }
  else if (ix.max_need > 3u) {  // terms that count four or five positive keys: hits as 3-bit numbers
    if (LA <= 8) { if (DT <= 8) KT_AGG_BM_CASE(8, 8, true, 5) else KT_AGG_BM_CASE(16, 8, true, 5) }
    else if (LA <= 16) { if (DT <= 8) KT_AGG_BM_CASE(8, 16, true, 5) else KT_AGG_BM_CASE(16, 16, true, 5) }
    else { if (DT <= 8) KT_AGG_BM_CASE(8, 32, true, 5) else KT_AGG_BM_CASE(16, 32, true, 5) }
  }
  else if (LA <= 8) { if (DT <= 8) KT_AGG_BM_CASE(8, 8, true, 3) else KT_AGG_BM_CASE(16, 8, true, 3) }
  else if (LA <= 16) { if (DT <= 8) KT_AGG_BM_CASE(8, 16, true, 3) else KT_AGG_BM_CASE(16, 16, true, 3) }
  else { if (DT <= 8) KT_AGG_BM_CASE(8, 32, true, 3) else KT_AGG_BM_CASE(16, 32, true, 3) }
#endif
  if (after_scan) after_scan();
  sc.launched_blocks = nb, sc.launched_packed = packed;
  if (packed && sc.defer_reduce) {
    // the caller goes on with kt_reduce_finalize_packed
  } else if (packed) {
    const uint32_t rb_ = (uint32_t)kRecTileUnits / (bm_args.pk.rec_bytes >> 3);  // records per block
    if (ix.bm_max_thr > 0)
      hipLaunchKernelGGL(kt_reduce_packed_slabs, dim3((ix.bm_max_thr + rb_ - 1u) / rb_, ix.n_chunks), dim3(kRecBlock), 0, s, slab,
                         ix.bm_chunks, ix.bm_rank_t, nb, pods.D, bm_args.pk, sc.slab_tag, sc.epoch, ix.n_chunks > 1 ? 1 : 0, partial);
  } else {
    const uint32_t max_pieces = ix.bm_max_thr * (agg_rec_bytes(pods.D, sc.counts) / 16u);
    if (max_pieces > 0)
      hipLaunchKernelGGL(kt_reduce_bitmap_slabs, dim3((max_pieces + 63) / 64, ix.n_chunks, kSlabSplits), dim3(256), 0, s, slab,
                         ix.bm_chunks, ix.bm_rank_t, nb, pods.D, sc.counts ? 1 : 0, sc.sign, sc.slab_tag, sc.epoch, partial);
  }
  return packed ? (ix.n_chunks == 1 ? "kt_aggregate_bitmap_packed" : "kt_aggregate_bitmap_packed_chunked")
                : (ix.n_chunks == 1 ? "kt_aggregate_bitmap" : "kt_aggregate_bitmap_chunked");
}

}  // namespace kt
