// kt_kernels_check.hip — kt_check_bitmap: PreFilter for n pods through the exact term bitmaps of the selector index (gfx950).
#include <type_traits>

#include "kt_scan.h"

#ifdef KT_PROFILE_PHASES
// Development aid (tools/prof_phases.py, -DKT_PROFILE_PHASES builds only): cycles per phase, summed over the waves of every launch
__device__ unsigned long long kt_prof_check[16];
extern "C" int kt_debug_prof_check(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(kt_prof_check), 16 * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(kt_prof_check), z, 16 * 8); }
  return 0;
}
#define KT_PROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define KT_PROF_ADD(slot, dt) prof_acc[slot] += (dt)
#else
#define KT_PROF_T(var)
#define KT_PROF_ADD(slot, dt)
#endif

namespace kt {

// ---------------------------------------------------------------------------------------------------
// kt_check_bitmap — PreFilter for n pods (plugin.go:148-215): CheckThrottled of both controllers
// (throttle_controller.go:349-397, clusterthrottle_controller.go:378-425) + CheckThrottledFor
// (throttle_types.go:128-153, clusterthrottle_types.go:30-55) through the CheckRec algebra.
// WAVE-AUTONOMOUS: after a chunk of the index is staged, every wave walks its own 64-pod tiles and never meets a
// workgroup barrier until the next chunk — all of a tile's state belongs to the wave that owns the 64 pods.
//   lane = pod : the pod's record arrives as one 8-byte meta word + its atom row (LA/8 128-bit loads); scan_tile
//                (kt_scan.h) yields the pod's matched terms.  The term's throttle and the pod-independent part of its
//                verdict sit in one 8-byte LDS word (TermInfo, staged per chunk from the CheckRec flags): unless the
//                throttle is `tight` (kRecTight: some pod could exceed thr[] / head[]) the status follows from it and
//                the pod's non-zero mask — no request row, no threshold read — and is counted in the lane's registers.
//   lane = (tight match) : the few matches that need the comparison are listed per wave and drained with
//                lane = match: request row + thr[] / head[] gathered as 16-byte pieces, verdict back to the pod's
//                counters through LDS.
//   lane = pod : the 8-byte summary word.
// ---------------------------------------------------------------------------------------------------

constexpr uint32_t kListCap = 128;  // tight-match list entries per wave (512 B)
// Per 64-bit word of term numbers, rebuilt per launch from the CheckRec flags (lean instantiation: the PreFilter sweep):
// the pod-independent part of every term's verdict as masks, so that a lane settles ALL matches of a word that need no
// comparison with a handful of mask operations and three popcounts — only matches of `tight` throttles are peeled.
//   seg_lo / seg_hi : lowest / highest number of every run of one throttle's terms (a throttle with several terms is
//                     reported once: (v ^ (v - seg_lo)) & v with v = x | seg_hi keeps the lowest match of every run; the
//                     index never lets a run straddle a word)
template <int DT>
struct alignas(16) WordVerdict {
  uint64_t seg_lo, seg_hi;
  uint64_t tight;      // kRecTight: the request row has to be compared with thr[] / head[]
  uint64_t exc;        // kRecExceedsByCount
  uint64_t act;        // kRecActiveByCount (also folded into act_nib[0][*])
  uint64_t ins;        // kRecInsufficientByCount
  // "a pod that requests dimension d is `active`" (active_mask bit d), tabulated per NIBBLE of the pod's non-zero mask:
  // act_nib[q][v] = act | OR of the masks of the dimensions 4q + b with bit b of v set — a lane gets its pod's `active`
  // terms of the word with DT / 4 eight-byte reads and as many ORs (round 3 read DT masks and selected them one by one:
  // 4 b128 reads, 24 VALU and 16 registers per visited word at DT = 8)
  uint64_t act_nib[DT / 4][16];
};
// TermInfo word 0: throttle row | kTiAdj
//          word 1: active_mask (16 bits) | counter shift when the pod requests an active dimension << 16
//                  | counter shift otherwise << 22 | kTiTight
// counter shifts: 4 = exceeds, 24 = active, 44 = insufficient, 0 = not throttled (nothing to count)
constexpr uint32_t kTiAdj = 0x00100000u, kTiTight = 0x10000000u;
constexpr int kTiShActive = 16, kTiShIdle = 22;

// TermInfo + (WORDWISE) the WordVerdict masks of ONE term number c: the 64 lanes of a wave hold the 64 numbers of word c / 64
// (c = 64 * word + lane), the masks are ballots.  tinfo / wv: where the tables go — LDS in the scan kernels' chunk prologue,
// global memory in kt_build_verdict_images.
template <int DT, bool WORDWISE, class TI, class WV>
__device__ __forceinline__ void build_term_verdicts(uint32_t tt, uint32_t c, const u32x2* g_rflags, TI* tinfo, WV* wv) {
  const bool real = (tt & kTermReal) != 0;
  const uint32_t t = tt & kTermRowMask;
  const u32x2 rf = real ? g_rflags[t] : u32x2{0u, 0u};  // {flags, active_mask}
  u32x2 ti = {0u, 0u};
  if (real) {
    const uint32_t sh_act = (rf.x & kRecExceedsByCount) ? 4u : 24u;
    const uint32_t sh_idle = (rf.x & kRecExceedsByCount) ? 4u : (rf.x & kRecActiveByCount) ? 24u : (rf.x & kRecInsufficientByCount) ? 44u : 0u;
    ti.x = t | ((tt & kTermAdj) ? kTiAdj : 0u);
    ti.y = (rf.y & 0xFFFFu) | sh_act << kTiShActive | sh_idle << kTiShIdle | ((rf.x & kRecTight) ? kTiTight : 0u);
  }
  tinfo[c] = ti;
  if (WORDWISE) {  // this wave holds the 64 numbers of word c / 64: their verdict masks by ballot
    const uint32_t ln = c & 63u;
    const uint32_t tp = (uint32_t)__shfl_up((int)tt, 1), tn = (uint32_t)__shfl_down((int)tt, 1);
    const bool adj = (tt & kTermAdj) != 0;
    const bool same_prev = ln > 0u && adj && (tp & kTermReal) && (tp & kTermAdj) && (tp & kTermRowMask) == t;
    const bool same_next = ln < 63u && adj && (tn & kTermReal) && (tn & kTermAdj) && (tn & kTermRowMask) == t;
    const bool xc = (rf.x & kRecExceedsByCount) != 0;
    const uint64_t m_lo = __ballot(real && !same_prev), m_hi = __ballot(real && !same_next);
    const uint64_t m_tight = __ballot(real && (rf.x & kRecTight) != 0), m_exc = __ballot(real && xc);
    const uint64_t m_act = __ballot(real && (rf.x & kRecActiveByCount) != 0), m_ins = __ballot(real && (rf.x & kRecInsufficientByCount) != 0);
    WV* wvp = wv + (c >> 6);
    const uint64_t mine = ln == 0u ? m_lo : ln == 1u ? m_hi : ln == 2u ? m_tight : ln == 3u ? m_exc : ln == 4u ? m_act : m_ins;
    // act_nib[q][v], entry e = 16 q + v: lane e (and lane e - 64 ... : DT = 16 has 64 entries, one per lane)
    const uint32_t eq = ln >> 4, ev = ln & 15u;
    uint64_t tab = m_act;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const uint64_t m = __ballot(real && ((rf.y >> d) & 1u) != 0);
      tab |= ((uint32_t)(d >> 2) == eq && ((ev >> (d & 3)) & 1u)) ? m : 0ull;
    }
    typedef decltype(&wvp->seg_lo) u64p;  // (pointer to uint64_t in the address space of wv)
    if (ln < 6u) ((u64p)wvp)[ln] = mine;
    if (ln < 4u * (uint32_t)DT) ((u64p)wvp)[6u + ln] = tab;
  }
}
template <int DT>
__host__ __device__ constexpr uint32_t verdict_image_word_bytes() { return 64u * 8u; }  // TermInfo block; the WordVerdicts follow ALL TermInfo blocks

// kt_build_verdict_images — TermInfo + WordVerdict of every word of the index, in global memory: grid = (word groups, chunks),
// a wave per word.  Layout: TermInfo [total_words][64] (8 bytes each, global word order), then WordVerdict<DT> [total_words].
template <int DT>
__global__ __launch_bounds__(256) void kt_build_verdict_images(const unsigned char* blob, const BmChunk* chunks, const void* recs, int T,
                                                               uint32_t total_words, unsigned char* out) {
  const BmChunk ch = chunks[blockIdx.y];
  const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (w >= ch.n_words) return;  // wave-uniform
  const u32x2* g_rflags = (const u32x2*)rec_flags<DT>((void*)recs, T);
  const uint32_t* term_t = (const uint32_t*)(blob + ch.img_off + ch.off_term_t);
  const uint32_t c_local = w * 64u + (threadIdx.x & 63u);
  const uint32_t c_glob = (ch.w0 + w) * 64u + (threadIdx.x & 63u);
  build_term_verdicts<DT, true>(term_t[c_local], c_glob, g_rflags, (u32x2*)out, (WordVerdict<DT>*)(out + (size_t)total_words * 64u * 8u));
}
size_t verdict_images_bytes(uint32_t total_words, int D) {
  return (size_t)total_words * (64u * 8u + (D <= 8 ? sizeof(WordVerdict<8>) : sizeof(WordVerdict<16>))) + 64;
}
void launch_build_verdict_images(const IndexDev& ix, uint32_t total_words, const void* recs, int T, int D, void* out, hipStream_t s) {
  if (!ix.n_chunks || !ix.bm_max_words) return;
  const dim3 g_((ix.bm_max_words + 3u) / 4u, ix.n_chunks), b_(256);
  if (dt_bucket_ix(D) <= 8) hipLaunchKernelGGL(kt_build_verdict_images<8>, g_, b_, 0, s, ix.bm_blob, ix.bm_chunks, recs, T, total_words, (unsigned char*)out);
  else hipLaunchKernelGGL(kt_build_verdict_images<16>, g_, b_, 0, s, ix.bm_blob, ix.bm_chunks, recs, T, total_words, (unsigned char*)out);
}

// what a tile needs of its 64 pods before it can start (kt_check_bitmap's fetch_tile)
template <int LA, int NV = 1>
struct TileRec {
  uint32_t p;                  // pod row
  uint64_t meta;
  u32x4 raw[LA / 8];           // atom row
  unsigned long long carried;  // class counters of the earlier chunks
  int64_t v[NV];               // AGG (kt_sweep): the request row — ResourceAmountOfPod of the reconcile half
};
// (Requesting a tile's records early — before the chunk is staged, or behind the previous tile's scan — was measured in rounds 4
//  and 5 and not kept: the kernels are bound by instruction issue, not by the chain of trips to memory, and the second record
//  costs registers.  The forms are in the git history of this file.)

struct BmCheckArgs {
  const uint64_t* meta;  // pod tables
  const uint16_t* latom;
  const int64_t* req;
  const uint32_t* lpair;  // raw labels: slow paths only
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
  uint32_t off_cnt, off_list, off_tinfo, off_wv;
  uint32_t off_next;  // namespace-ordered sweeps: the workgroup's next tile (its waves take tiles as they get free)
  uint32_t n_slow;
  int32_t DS, LS, T;
  // small launches (SMALL instantiation: one workgroup per (chunk, tile), results met by atomics)
  uint32_t* ticket;        // [tiles] arrival counters, zero between launches
  uint64_t* host_summary;  // nullable: pinned host copy of the final summary words
  uint32_t has_overflow;   // some pod carries more relevant atoms than its atom row holds (kMetaOverflow)
  // namespace-ordered sweeps (ix.by_ns): record j of the views belongs to pod rows[j]
  const uint64_t* v_meta;
  const uint16_t* v_latom;
  uint64_t* carry;         // [n] class counters between chunks
  const uint32_t* wg_range;  // nullable: record range of every workgroup, cut at namespace boundaries (plan_wg_ranges, host side)
  uint32_t n_inline;       // > 0: the pod rows travel in the argument block (no staging copy)
  int64_t inline_rows[8];
  // namespace-ordered lean sweeps of multi-chunk programs: TermInfo + WordVerdict of EVERY word, built once per generation
  // of CheckRecs by kt_build_verdict_images (below) — the chunk prologue then is a straight copy into LDS (nullable: the
  // other forms build their chunk's tables in place)
  const unsigned char* wv_img;
  uint32_t wv_total_words;
  // AGG instantiation (kt_sweep_launch: the PreFilter sweep AND the reconcile scan in one pass over the pod rows): the
  // packed fold of kt_aggregate_bitmap (kt_kernels_aggregate.hip) rides on the check's scan
  unsigned char* slab;
  uint32_t* slab_tag;
  uint32_t epoch;
  uint32_t off_rank, off_tab;
  uint32_t pk_nw, pk_rec;  // PackPlan::nw / rec_bytes
  uint32_t pk_dim[8];      // per dimension: shift | pos << 8 | word << 16 | (width != 0) << 24
};

static BmCheckArgs make_bm_check_args(const PodTable& pods, int64_t n, const int64_t* rows, const SelProgram& sp,
                                      const SelProgram* sp_dev, const IndexDev& ix, const void* recs,
                                      uint64_t* summary, uint8_t* status, uint32_t* total, const PackPlan* agg_pk = nullptr) {
  BmCheckArgs a{};
  a.meta = pods.meta, a.latom = pods.latom, a.req = pods.req, a.lpair = pods.lpair, a.lkey = pods.lkey;
  a.n = n, a.rows = rows, a.recs = recs, a.summary = summary, a.status = status;
  a.sp = sp_dev, a.ns_valid = sp.ns_valid, a.slow_thr = ix.slow_thr, a.n_slow = ix.n_slow;
  a.DS = pods.DS, a.LS = pods.LS, a.T = sp.T;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  a.off_cnt = take(kBlockIx * 8);
  a.off_list = take((kBlockIx / kWave) * kListCap * 4);
  a.off_next = take(16);
  a.off_tinfo = take(ix.bm_max_words * 64u * 8u);
  a.off_wv = take(ix.bm_max_words * (uint32_t)(pods.D <= 8 ? sizeof(WordVerdict<8>) : sizeof(WordVerdict<16>)));
  plan_bitmap_index(ix, a.ix, take);
  if (agg_pk) {  // the reconcile half's tables: ranks of the term numbers, one packed record per throttle of the chunk
    a.pk_nw = agg_pk->nw, a.pk_rec = agg_pk->rec_bytes;
    for (int d = 0; d < 8; ++d)
      a.pk_dim[d] = (uint32_t)agg_pk->shift[d] | (uint32_t)agg_pk->pos[d] << 8 | (uint32_t)agg_pk->word[d] << 16 | (agg_pk->width[d] ? 1u << 24 : 0u);
    a.off_rank = take(ix.bm_max_words * 64u * 2u);
    a.off_tab = take(ix.bm_max_thr * agg_pk->rec_bytes);
  }
  *total = o;
  return a;
}

int check_sweep_blocks(int64_t n) {
  const int64_t nb = (n + kBlockIx - 1) / kBlockIx;
  return (int)(nb > kCUs ? kCUs : nb < 1 ? 1 : nb);
}
uint32_t check_fixed_lds() { return kBlockIx * 8 + (kBlockIx / kWave) * kListCap * 4 + 64; }
static_assert(sizeof(WordVerdict<16>) == 560 && 64u * 8u + sizeof(WordVerdict<16>) == kCheckWordLds, "kCheckWordLds (kt_index.h) follows WordVerdict");
static_assert(sizeof(WordVerdict<8>) == 304, "WordVerdict<8>");
// TermInfo + WordVerdict per 64-bit word, for an engine with D dimensions
uint32_t check_word_lds(int D) { return 64u * 8u + (uint32_t)(D <= 8 ? sizeof(WordVerdict<8>) : sizeof(WordVerdict<16>)); }

// SMALL: a launch of a few pods (one PreFilter call, an admission queue): grid = (chunks, tiles), every workgroup scans
//        ONE tile against ONE chunk, so that the chunks of a large program are walked side by side instead of one after
//        the other; the class counters meet in the summary word by atomics, the last workgroup of a tile (arrival
//        ticket) writes the word's final form — also to a pinned host copy, which saves the D2H copy of the fetch.
// WPE:  waves per SIMD the register allocation has to leave room for (4: one workgroup per CU, 8: two)
// FULL: the launch also wants the status matrix and / or has throttles on the slow list — the lean instantiation
//       (summary words only, no slow list: the PreFilter sweep) keeps neither code path nor their registers
// ONE:  the index is ONE chunk and the sweep runs in row order (the two-per-CU sweep of programs that fit half the LDS:
//       BASELINE configs 1-3) — no chunk loop, no carry words, no namespace-ordered views: none of their registers either
//       (the 64-VGPR instantiation spilled a handful of them, and a reload in the chunk prologue is a trip to memory)
// AGG:  kt_sweep — the ONE lean sweep also IS the reconcile scan: a lane whose pod is counted (shouldCountIn,
//       throttle_controller.go:217-219) folds its packed request words (PackPlan) into the record of every throttle it
//       matches, exactly as kt_aggregate_bitmap's packed instantiation does, and the workgroup spills its table of records
//       as a slab for kt_reduce_finalize_packed.  One pass over the pod rows, one chunk prologue, one scan per pod where
//       check + aggregate made two; the check's verdicts are those against the status stored BEFORE this reconcile.
template <int DT, int LA, bool VETO, int NEED, int WPE, bool FULL, bool SMALL, bool ONE = false, bool AGG = false>
__global__ __launch_bounds__(kBlockIx, WPE) void kt_check_bitmap(const BmCheckArgs a) {
  static_assert(!AGG || (ONE && !FULL && !SMALL && WPE < 8), "the fused sweep is the lean single-chunk form, one workgroup per CU");
  // (pieces of the three rows per batch: 12 registers each — the whole rows of a 16-dimension engine would be 96 registers
  //  of a 128-register kernel: four pieces per batch at most, two batches there)
  constexpr int kDrainUnroll = WPE >= 8 ? 1 : (DT / 2 > 4 ? 4 : DT / 2);  // drains in the middle of a scan (the list ran full)
  // the drain behind the scan (two pieces per batch in the 64-VGPR form: the whole rows at once cost it 80 B of scratch)
  constexpr int kDrainFinalUnroll = WPE >= 8 ? 2 : (DT / 2 > 4 ? 4 : DT / 2);
  // WORDWISE: matches that need no comparison are settled per 64-bit word with mask algebra (WordVerdict) instead of
  // being peeled one by one: three popcounts per visited word, and only the matches of tight throttles go through the
  // peel.  Round 3 had it in the one-workgroup-per-CU instantiation only (config 4's 130 matches per pod: 1.27 -> 0.92 ms):
  // with one mask per dimension its registers did not fit the 64-VGPR budget of the two-per-CU one (config 2: 33.6 -> 38.2
  // us, 56 B of scratch).  With the per-nibble tables (WordVerdict::act_nib) a visit holds 12 mask registers instead of
  // 28, and the two-per-CU instantiation takes it too: config 2's peel was 10.7 steps per tile at 21 % busy lanes,
  // half of the kernel's instructions.
  constexpr bool WORDWISE = !FULL;
  // the verdict masks of a word are requested together with its atom rows (scan_tile's pre hook) where registers allow;
  // the 64-VGPR instantiation reads them when the rows have been consumed
  constexpr bool PREFETCH = WPE < 8;
#ifdef KT_PROFILE_PHASES
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  KT_PROF_T(t_kernel0);
#endif
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  const CheckRec<DT>* recs = (const CheckRec<DT>*)a.recs;
  const u32x2* g_rflags = (const u32x2*)rec_flags<DT>((void*)a.recs, a.T);
  const uint32_t n = (uint32_t)a.n;  // pod_capacity <= 2^31
  const int DS = a.DS;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: LDS bases stay scalar
  // this wave's private LDS areas
  lds_u64wp cnt = (lds_u64wp)(lds + a.off_cnt) + wave * kWave;       // [64] class counters coming back from the drain
  lds_u32wp list = (lds_u32wp)(lds + a.off_list) + wave * kListCap;  // tight matches: lane << 20 | throttle row
  KT_LDS u32x2* tinfo = (KT_LDS u32x2*)(lds + a.off_tinfo);
  const uint32_t n_wtiles = (n + kWave - 1) / kWave;
  const uint32_t wstep = gridDim.x * (kBlockIx / kWave);
  const uint32_t n_chunks = ONE ? 1u : a.ix.n_chunks;
  const uint32_t c_lo = SMALL ? blockIdx.x : 0u, c_hi = SMALL ? blockIdx.x + 1u : n_chunks;
  // namespace order (a.ix.by_ns; rows[] sorted by namespace, results indexed by pod row): this workgroup owns the
  // tiles [t_lo, t_hi) and only walks the chunks that hold words of their namespaces
  const bool by_ns = !SMALL && !ONE && a.ix.by_ns != 0u;
  uint32_t t_lo = 0, t_hi = n_wtiles, ns_lo = 0, ns_hi = 0, first_ci = 0u, last_ci = n_chunks - 1u;
  // the records the workgroup's tiles are cut from: [rec0, rec_end) — tile wt holds records rec0 + 64 wt .. (everything, or
  // this workgroup's range of a namespace-ordered list: planned ranges end at namespace boundaries where one lies close)
  uint32_t rec0 = 0, rec_end = n;
  if (by_ns) {
    if (a.wg_range) {
      rec0 = __builtin_amdgcn_readfirstlane(a.wg_range[blockIdx.x]), rec_end = __builtin_amdgcn_readfirstlane(a.wg_range[blockIdx.x + 1u]);
      if (rec0 >= rec_end) return;
      t_lo = 0u, t_hi = (rec_end - rec0 + kWave - 1u) / kWave;
    } else {
      const uint32_t tpb = (n_wtiles + gridDim.x - 1u) / gridDim.x;
      t_lo = min(blockIdx.x * tpb, n_wtiles), t_hi = min(t_lo + tpb, n_wtiles);
      if (t_lo >= t_hi) return;
    }
    ns_lo = (uint32_t)(a.v_meta[(uint64_t)rec0 + (uint64_t)t_lo * kWave] & kMetaNsMask);
    ns_hi = (uint32_t)(a.v_meta[min((uint64_t)rec0 + (uint64_t)t_hi * kWave, (uint64_t)rec_end) - 1u] & kMetaNsMask);
    ns_lo = __builtin_amdgcn_readfirstlane(ns_lo), ns_hi = __builtin_amdgcn_readfirstlane(max(ns_hi, ns_lo));
    relevant_chunk_span(a.ix, ns_lo, ns_hi, first_ci, last_ci);
    last_ci = max(last_ci, first_ci);
  }
  for (uint32_t ci = SMALL ? c_lo : first_ci; ci < (SMALL ? c_hi : last_ci + 1u); ++ci) {
    if (by_ns && ci != first_ci && !chunk_relevant(a.ix, ci, ns_lo, ns_hi)) continue;
    const bool first = ONE || ci == (SMALL ? 0u : first_ci), last = ONE || ci == last_ci;
    const BmChunk ch = a.ix.chunks[ci];
    // the tile's records, always from valid addresses: lanes past the end re-read the last pod and are switched off by
    // `on`
    const uint32_t wt0 = SMALL ? (wave == 0 ? blockIdx.y : n_wtiles) : by_ns ? t_lo + wave : blockIdx.x * (kBlockIx / kWave) + wave;
    const uint32_t wt_step = SMALL ? n_wtiles : by_ns ? (uint32_t)(kBlockIx / kWave) : wstep;
    auto fetch_tile = [&](uint32_t wt) {
      TileRec<LA, AGG ? DT : 1> r;
      const uint32_t i = rec0 + wt * kWave + lane;
      const uint32_t ic = min(i, rec_end - 1u);
      r.p = (SMALL && a.n_inline) ? (uint32_t)a.inline_rows[ic & 7u] : a.rows ? (uint32_t)a.rows[ic] : ic;
      r.meta = by_ns ? a.v_meta[ic] : a.meta[r.p];
      load_atoms<LA>(by_ns ? a.v_latom : a.latom, by_ns ? ic : r.p, r.raw);
      const unsigned long long* carry_w = by_ns ? (const unsigned long long*)a.carry + ic : (const unsigned long long*)a.summary + (by_ns ? r.p : i);
      r.carried = (!SMALL && !first && i < rec_end) ? __hip_atomic_load(carry_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      if constexpr (AGG) load_requests<DT>(a.req, DS, (int64_t)r.p, r.v);  // every lane's: the row does not hang off the meta word
      return r;
    };
    TileRec<LA, AGG ? DT : 1> cur{};
    KT_PROF_T(t_b0);
    __syncthreads();  // nobody reads the previous image any more
    KT_PROF_T(t_b1);
    KT_PROF_ADD(0, t_b1 - t_b0);
    const BmView bm = open_chunk<VETO>(lds, a.ix, ch);
    if (by_ns && threadIdx.x == 0) *(lds_u32wp)(lds + a.off_next) = t_lo;
    const bool from_images = !SMALL && !ONE && !FULL && a.wv_img != nullptr;  // (kernel argument: uniform)
    // (tables built in place: the first term word of every thread is requested AHEAD of the image — it arrives with that batch,
    //  and the dependent read of the CheckRec flags is the prologue's second trip to memory instead of its third)
    const uint32_t* term_t = (const uint32_t*)(a.ix.blob + ch.img_off + ch.off_term_t);
    uint32_t tt_first = 0u;
    if (!from_images && threadIdx.x < ch.n_words * 64u) tt_first = term_t[threadIdx.x];
    {  // everything that is a straight copy into LDS goes out as ONE batch of loads: the image, the fold's ranks (AGG), and
       // — sweeps of multi-chunk programs — the chunk's words of the TermInfo / WordVerdict tables kt_build_verdict_images left
      const u32x4* img0 = (const u32x4*)(a.ix.blob + ch.img_off);
      StageSeg segs[4] = {chunk_image_segment(a.ix, ch), StageSeg{0u, img0, 0u}, StageSeg{0u, img0, 0u}, StageSeg{0u, img0, 0u}};
      if constexpr (AGG) segs[1] = StageSeg{a.off_rank, (const u32x4*)(a.ix.blob + ch.img_off + ch.off_term_rank), ch.n_words * 8u};
      if (from_images) {
        segs[2] = StageSeg{a.off_tinfo, (const u32x4*)(a.wv_img + (size_t)ch.w0 * verdict_image_word_bytes<DT>()), ch.n_words * (64u * 8u / 16u)};
        if (WORDWISE)
          segs[3] = StageSeg{a.off_wv, (const u32x4*)(a.wv_img + (size_t)a.wv_total_words * 64u * 8u + (size_t)ch.w0 * sizeof(WordVerdict<DT>)),
                             ch.n_words * (uint32_t)(sizeof(WordVerdict<DT>) / 16u)};
      }
      lds_stage_segments<4>(lds, segs);
    }
    if constexpr (AGG) {  // the reconcile half: the table of packed records zeroed
      const uint32_t tab_bytes = (ch.n_thr * a.pk_rec + 15u) & ~15u;
      for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += kBlockIx) ((lds_u32wp)(lds + a.off_tab))[i] = 0u;
    }
    if (from_images) {
      // (the tables of this chunk's words came with the image: kt_build_verdict_images built them once per generation of CheckRecs)
    } else {
      // TermInfo of the chunk's term numbers: throttle row + the pod-independent verdict bits of its CheckRec
      for (uint32_t c = threadIdx.x; c < ch.n_words * 64u; c += kBlockIx)
        build_term_verdicts<DT, WORDWISE>(c == threadIdx.x ? tt_first : term_t[c], c, g_rflags, tinfo, (KT_LDS WordVerdict<DT>*)(lds + a.off_wv));
    }
    KT_PROF_T(t_b2);
    __syncthreads();
    KT_PROF_T(t_b3);
    KT_PROF_ADD(1, t_b2 - t_b1);
    KT_PROF_ADD(2, t_b3 - t_b2);
    KT_PROF_ADD(6, 1ull);
    // namespace-ordered sweeps: the waves of the workgroup take the tiles of its range as they get free (a counter in LDS) —
    // with a fixed stride every chunk pass ended with the waves that own five tiles while those that own four waited
    auto next_tile = [&](uint32_t prev) -> uint32_t {
      if (by_ns) {
        uint32_t t = 0u;
        if (lane == 0u) t = lds_add((lds_u32wp)(lds + a.off_next), 1u);
        return __builtin_amdgcn_readfirstlane(t);
      }
      return prev + wt_step;
    };
    uint32_t wt = by_ns ? next_tile(0u) : wt0;
    uint32_t wt_next = 0u;
    for (; wt < t_hi; wt = wt_next) {
      // ---- the tile's records
      KT_PROF_T(t_f0);
      cur = fetch_tile(wt);
#ifdef KT_PROFILE_PHASES
      __builtin_amdgcn_s_waitcnt(0);  // (vmcnt / lgkmcnt / expcnt = 0: the wait for the records is charged to the fetch)
      KT_PROF_T(t_f1);
      KT_PROF_ADD(3, t_f1 - t_f0);
      KT_PROF_ADD(7, 1ull);
#endif
      const uint32_t i = rec0 + wt * kWave + lane;
      const bool in = i < rec_end;
      const uint32_t ic = min(i, rec_end - 1u);
      const uint32_t p = cur.p;
      const uint32_t si = by_ns ? p : i;  // the pod's index in the summary words / status matrix
      const uint64_t meta = cur.meta;
      u32x4 raw[LA / 8];
#pragma unroll
      for (int q = 0; q < LA / 8; ++q) raw[q] = cur.raw[q];
      // class counters so far (bit 1 = error) ride in the summary word (namespace order: the carry word) between chunks
      unsigned long long* carry_w = by_ns ? (unsigned long long*)a.carry + ic : (unsigned long long*)a.summary + si;
      const unsigned long long carried = cur.carried;
      cnt[lane] = 0ull;
      unsigned long long my = carried & ~3ull;  // this lane's class counters
      const bool on = in && ((meta >> kMetaStateShift) & kPodValid) != 0;
      const uint32_t ns = on ? (uint32_t)(meta & kMetaNsMask) : 0u;
      const uint32_t nz = (uint32_t)(meta >> kMetaNzShift);
      // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
      // (the byte is requested here and looked at behind the scan: a tile is a chain of dependent trips to memory — two
      //  tiles per wave at 1M pods — and this one hangs off the record's namespace)
      const uint32_t ns_ok_raw = a.ns_valid[ns];
      bool pod_err = (carried & 2ull) != 0;
      uint32_t ro[LA];
      atom_row_offsets<LA>(raw, ro);
      uint32_t n_list = 0;  // wave-uniform
      uint32_t last_t = 0xFFFFFFFFu;
      // AGG: is the pod counted (shouldCountIn && isNotFinished: throttle_controller.go:217-219, pod_util.go:26-28), and
      // ResourceAmountOfPod as the packed words of the fold — what kt_build_scan_view stores per record of the aggregate's
      // view, built here from the request row: pod count 1 from bit 0 of word 0, every request as its field
      bool counted = false;
      uint32_t zero_keys = 0u;
      unsigned long long pw[4] = {0ull, 0ull, 0ull, 0ull};
      if constexpr (AGG) {
        const uint32_t st = (uint32_t)(meta >> kMetaStateShift) & 0xFu;
        counted = in && (st & (kPodValid | kPodSchedMatch | kPodScheduled | kPodFinished)) == (kPodValid | kPodSchedMatch | kPodScheduled);
        const uint32_t present = (uint32_t)(meta >> kMetaPresentShift) & 0xFFFFu;
        zero_keys = present & ~nz & 0xFFFFu;  // keys carried with the value 0
        pw[0] = 1ull;
        static_assert(!AGG || DT == 8, "pk_dim holds eight dimensions");
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const uint32_t pd = a.pk_dim[d];  // (wave-uniform: scalar decode)
          const uint64_t f = (uint64_t)cur.v[d] >> (pd & 63u);
          const uint64_t piece = (pd >> 24) ? f << ((pd >> 8) & 63u) : 0ull;  // (dimensions without a field: width 0)
          const uint32_t k = (pd >> 16) & 3u;
          pw[0] |= k == 0u ? piece : 0ull, pw[1] |= k == 1u ? piece : 0ull, pw[2] |= k == 2u ? piece : 0ull, pw[3] |= k == 3u ? piece : 0ull;
        }
      }

      // UNROLL (a std::integral_constant): pieces of the request row / thr[] / head[] requested per batch.  The drain behind
      // the scan (nothing of the scan is live any more) asks for the whole rows at once in every instantiation: with two
      // tiles per wave the kernel is a latency chain, and the 64-VGPR instantiation's one-piece-at-a-time loop was four
      // dependent trips to memory per tile
      auto drain = [&](auto unroll_tag) {
        constexpr int UNROLL = decltype(unroll_tag)::value;
        // ---- lane = listed (pod lane, throttle): the full comparison; pod row / non-zero mask come from the pod's
        //      lane by ds_bpermute
        for (uint32_t base = 0; base < n_list; base += kWave) {
          const uint32_t j = base + lane;
          const bool vv = j < n_list;
          const uint32_t e = list[vv ? j : 0u];
          // the list holds throttle rows — or, WORDWISE, term numbers (their row is one LDS read away, taken here with
          // every lane busy instead of in the peel)
          const uint32_t pl = e >> 20;
          // WORDWISE: the term's TermInfo (LDS) has the throttle row AND the pod-independent verdict bits — no trip to the
          // flags in global memory before the rows can be requested
          u32x2 ti = {0u, 0u};
          if (WORDWISE) ti = tinfo[e & kTermRowMask];
          const uint32_t t = WORDWISE ? ti.x & kTermRowMask : e & kTermRowMask;
          const uint32_t prow = (uint32_t)__shfl((int)p, (int)pl);
          const uint32_t psi = FULL ? (uint32_t)__shfl((int)si, (int)pl) : 0u;
          const uint32_t pnz = (uint32_t)__shfl((int)nz, (int)pl);
          const CheckRec<DT>* rc = recs + t;
          const kt_i64x2* xr = (const kt_i64x2*)(a.req + (uint64_t)prow * (uint32_t)DS);
          u32x2 fa;  // {flags, active_mask}
          if (WORDWISE) {
            const uint32_t sh_act = (ti.y >> kTiShActive) & 63u, sh_idle = (ti.y >> kTiShIdle) & 63u;
            fa.x = (sh_act == 4u ? kRecExceedsByCount : 0u) | (sh_idle == 24u ? kRecActiveByCount : 0u) | (sh_idle == 44u ? kRecInsufficientByCount : 0u);
            fa.y = ti.y & 0xFFFFu;
          } else {
            fa = g_rflags[t];
          }
          bool exc = false, ins = false;
#pragma unroll 1
          for (int q0 = 0; q0 < DT / 2; q0 += UNROLL) {
            // one batch: UNROLL pieces of the three rows in flight together, nothing looks at a value before all are requested
            kt_i64x2 x[UNROLL], th[UNROLL], hd[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              const int q = q0 + u;
              x[u] = xr[2 * q < DS ? q : 0];  // pieces past the row re-read piece 0 and are masked by pnz
              th[u] = *(const kt_i64x2*)(rc->thr + 2 * q);
              hd[u] = *(const kt_i64x2*)(rc->head + 2 * q);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              const int q = q0 + u;
              const bool nz0 = (pnz >> (2 * q)) & 1u, nz1 = (pnz >> (2 * q + 1)) & 1u;
              exc |= (nz0 && x[u].x > th[u].x) || (nz1 && x[u].y > th[u].y);
              ins |= (nz0 && x[u].x > hd[u].x) || (nz1 && x[u].y > hd[u].y);
            }
          }
          exc |= (fa.x & kRecExceedsByCount) != 0, ins |= (fa.x & kRecInsufficientByCount) != 0;
          const bool act = (fa.x & kRecActiveByCount) || (pnz & fa.y);
          const uint32_t st = exc ? 4u : act ? 2u : ins ? 3u : 1u;
          if (vv) {
            if (st != 1u) lds_add64(cnt + pl, st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : 1ull << 44);
            if (FULL && a.status) a.status[(uint64_t)psi * (uint32_t)a.T + t] = (uint8_t)st;
          }
        }
        n_list = 0;
      };
      auto push = [&](bool want, uint32_t t) {  // wave-wide append of (lane, t)
        const uint64_t mk = __ballot(want);
        if (mk == 0ull) return;
        if (want) list[n_list + lane_rank(mk)] = lane << 20 | t;
        n_list += (uint32_t)__popcll(mk);
        if (n_list > kListCap - kWave) drain(std::integral_constant<int, kDrainUnroll>());
      };

      // a pod whose relevant atoms did not fit its atom row is not scanned through the bitmaps: its lane walks EVERY
      // throttle's terms over the raw labels instead (with the first chunk)
      const bool overflow = FULL && a.has_overflow && on && (meta & kMetaOverflow) != 0;
      const bool scan_on = on && !overflow;
      // ---- throttles with unconvertible selectors: walked once, with the first chunk (error semantics depend on
      //      term order, throttle_selector.go:30-42); their matches take the full comparison
      if (FULL && first && (a.n_slow || a.has_overflow)) {
        const SelProgram& sp = *a.sp;
        const uint32_t* lp = a.lpair + (uint64_t)p * (uint32_t)a.LS;
        const uint32_t* lk = a.lkey + (uint64_t)p * (uint32_t)a.LS;
        for (uint32_t ks = 0; ks < a.n_slow; ++ks) {
          const int ts = (int)a.slow_thr[ks];
          const uint32_t res = walk_slow_mem(sp, ts, sp.ns_term_ok + (size_t)ns * sp.gw, scan_on, lp, lk, a.LS);
          if (res & kSlowError) pod_err = true;
          push((res & kSlowMatched) && scan_on, (uint32_t)ts);
        }
        if (__ballot(overflow) != 0ull)
          for (int t = 0; t < a.T; ++t) {
            const uint32_t res = walk_slow_mem(sp, t, sp.ns_term_ok + (size_t)ns * sp.gw, overflow, lp, lk, a.LS);
            if (res & kSlowError) pod_err = true;
            push((res & kSlowMatched) && overflow, (uint32_t)t);
          }
      }

      if (!WORDWISE) {
      scan_tile<LA, VETO, NEED, (VETO && WPE < 8)>(
          bm, scan_on, ns, ro,
          [&](bool has, uint32_t c) {
            // branch-free: the term's TermInfo word carries both verdicts a non-tight throttle can give (as the bit
            // position of the class counter to bump: 4 / 24 / 44, 0 = not throttled), chosen by the pod's non-zero mask
            const u32x2 ti = tinfo[c];
            const uint32_t t = ti.x & kTermRowMask;
            // a throttle with several terms is reported once
            const bool ok = has && !((ti.x & kTiAdj) && t == last_t);
            last_t = ok ? t : last_t;
            const bool tight = (ti.y & kTiTight) != 0;
            const uint32_t sh = (nz & ti.y & 0xFFFFu) ? (ti.y >> kTiShActive) & 63u : (ti.y >> kTiShIdle) & 63u;
            const bool fast = ok && !tight;
            const unsigned long long inc = 1ull << sh;
            my += (fast && sh != 0u) ? inc : 0ull;
            if (FULL && a.status && fast) a.status[(uint64_t)si * (uint32_t)a.T + t] = (uint8_t)(sh == 4u ? 4u : sh == 24u ? 2u : sh == 44u ? 3u : 1u);
            push(ok && tight, t);
          },
          [&](uint32_t c) {
            return term_match_mem(*a.sp, bm.term_g[c], a.lpair + (uint64_t)p * (uint32_t)a.LS, a.lkey + (uint64_t)p * (uint32_t)a.LS, a.LS);
          });
      } else {
      // every match that needs no comparison is settled per WORD with mask algebra (WordVerdict), the matches of tight
      // throttles are peeled into the list as bare term numbers
      KT_LDS const unsigned char* wv = lds + a.off_wv;
      // this pod's entries of the per-nibble `active` tables, as byte offsets into a word's WordVerdict (kept in registers
      // where there is room; the 64-VGPR instantiation recomputes them from the non-zero mask per visited word)
      auto nib_offset = [&](int q) { return (uint32_t)offsetof(WordVerdict<DT>, act_nib) + (uint32_t)q * 128u + ((nz >> (4 * q)) & 15u) * 8u; };
      uint32_t nib_off[DT / 4];
#pragma unroll
      for (int q = 0; q < DT / 4; ++q) nib_off[q] = PREFETCH ? nib_offset(q) : 0u;
      const bool seg_on = ch.has_adj != 0u;  // wave-uniform: some throttle of the chunk has several terms
      // class counters of this tile (v_bcnt accumulates); they start from what the earlier chunks carried over
      uint32_t n_exc = (uint32_t)(my >> 4) & 0xFFFFFu, n_act = (uint32_t)(my >> 24) & 0xFFFFFu, n_ins = (uint32_t)(my >> 44);
      struct VerdictRegs {
        u64x2 seg, te, ai;
        uint64_t nib[DT / 4];
      };
      auto fetch = [&](uint32_t w) -> VerdictRegs {
        KT_LDS const unsigned char* q = wv + __umul24(w, (uint32_t)sizeof(WordVerdict<DT>));
        VerdictRegs r;
        r.seg = u64x2{0ull, 0ull};
        if (seg_on) r.seg = *(KT_LDS const u64x2*)(q + offsetof(WordVerdict<DT>, seg_lo));  // {seg_lo, seg_hi}
        r.te = *(KT_LDS const u64x2*)(q + offsetof(WordVerdict<DT>, tight));              // {tight, exc}
        r.ai = *(KT_LDS const u64x2*)(q + offsetof(WordVerdict<DT>, act));                // {act, ins}
#pragma unroll
        for (int k = 0; k < DT / 4; ++k) r.nib[k] = *(KT_LDS const unsigned long long*)(q + (PREFETCH ? nib_off[k] : nib_offset(k)));
        return r;
      };
      // AGG: the fold of kt_aggregate_bitmap's packed instantiation over the word's matches (one per throttle: x is past
      // the run rule below) — counted pods only, lane = pod, one LDS atomic per packed word
      KT_LDS const uint16_t* trank = (KT_LDS const uint16_t*)(lds + a.off_rank);
      KT_LDS unsigned char* tab = lds + a.off_tab;
      const uint32_t pk_rec = AGG ? a.pk_rec : 0u, pk_nw = AGG ? a.pk_nw : 0u;
      auto fold = [&](uint32_t w, uint64_t x) {
        uint64_t xf = counted ? x : 0ull;
        auto add = [&](bool has, uint32_t r) {
          if (has) {
            KT_LDS unsigned char* rp = tab + __umul24(r, pk_rec);
            lds_u64wp tv = (lds_u64wp)rp;
            lds_add64(tv, pw[0]);
            if (pk_nw > 1u) lds_add64(tv + 1, pw[1]);
            if (pk_nw > 2u) lds_add64(tv + 2, pw[2]);
            if (pk_nw > 3u) lds_add64(tv + 3, pw[3]);
            if (zero_keys) (void)__hip_atomic_fetch_or((lds_u32wp)(rp + pk_nw * 8u), zero_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        };
        while (__ballot(xf != 0ull) != 0ull) {  // two matches per step: both rank reads in flight together
          const bool h1 = xf != 0ull;
          const uint32_t c1 = h1 ? w * 64u + (uint32_t)__ffsll((unsigned long long)xf) - 1u : 0u;
          xf &= xf - 1ull;
          const bool h2 = xf != 0ull;
          const uint32_t c2 = h2 ? w * 64u + (uint32_t)__ffsll((unsigned long long)xf) - 1u : 0u;
          xf &= xf - 1ull;
          const uint32_t r1 = trank[c1] & 0x7FFFu, r2 = trank[c2] & 0x7FFFu;  // chunk-local throttle ranks
          add(h1, r1);
          add(h2, r2);
        }
      };
      auto settle = [&](uint32_t w, uint64_t x, const VerdictRegs& q) -> uint64_t {
        if (seg_on) {  // a throttle with several terms is reported once: the lowest match of every run
          const uint64_t v = x | q.seg.y;
          x = andn_64(x, v - q.seg.x);  // (= x & (v ^ (v - seg_lo)) & v, x being part of v)
        }
        if constexpr (AGG) fold(w, x);
        uint64_t act = q.nib[0];  // (act-by-count is part of every entry of nibble 0's table)
#pragma unroll
        for (int k = 1; k < DT / 4; ++k) act |= q.nib[k];
        const uint64_t xe = and_not_and_64(x, q.te.x, q.te.y);  // not tight, exceeded by count
        const uint64_t xf = and_not_not_64(x, q.te.x, q.te.y);  // settled here, not exceeded by count
        const uint64_t xa = xf & act, xi = and_not_and_64(xf, act, q.ai.y);
        n_exc = (uint32_t)__popc((uint32_t)xe) + ((uint32_t)__popc((uint32_t)(xe >> 32)) + n_exc);
        n_act = (uint32_t)__popc((uint32_t)xa) + ((uint32_t)__popc((uint32_t)(xa >> 32)) + n_act);
        n_ins = (uint32_t)__popc((uint32_t)xi) + ((uint32_t)__popc((uint32_t)(xi >> 32)) + n_ins);
        return x & q.te.x;
      };
      auto confirm_slow = [&](uint32_t c) {
        return term_match_mem(*a.sp, bm.term_g[c], a.lpair + (uint64_t)p * (uint32_t)a.LS, a.lkey + (uint64_t)p * (uint32_t)a.LS, a.LS);
      };
      auto peel_tight = [&](bool has, uint32_t c) { push(has, c); };
      if constexpr (PREFETCH) {
        scan_tile<LA, VETO, NEED, VETO>(
            bm, scan_on, ns, ro, peel_tight, confirm_slow,
            [&](uint32_t w, uint64_t x, const VerdictRegs& q) -> uint64_t { return settle(w, x, q); },
            [&](uint32_t w) -> VerdictRegs { return fetch(w); });
      } else {
        scan_tile<LA, VETO, NEED, false>(
            bm, scan_on, ns, ro, peel_tight, confirm_slow,
            [&](uint32_t w, uint64_t x, int) -> uint64_t {
              if (__ballot(x != 0ull) == 0ull) return 0ull;  // nobody of the tile matched a term of this word
              return settle(w, x, fetch(w));
            });
      }
      my = (unsigned long long)n_exc << 4 | (unsigned long long)n_act << 24 | (unsigned long long)n_ins << 44;
      }
      wt_next = next_tile(wt);
#ifdef KT_PROFILE_PHASES
      KT_PROF_T(t_s1);
      KT_PROF_ADD(4, t_s1 - t_f1);
#endif
      if (n_list) drain(std::integral_constant<int, kDrainFinalUnroll>());
      pod_err |= on & (ns_ok_raw == 0u);
      // ---- lane = pod: the 8-byte summary word
      if (SMALL) {
        // this workgroup's share of the tile's counters; the last workgroup to arrive gives the words their final form
        if (in) {
          const unsigned long long c = my + cnt[lane];
          if (c) (void)__hip_atomic_fetch_add((unsigned long long*)a.summary + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pod_err) (void)__hip_atomic_fetch_or((unsigned long long*)a.summary + i, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        uint32_t arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(a.ticket + wt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived + 1u == n_chunks) {
          if (in) {
            const unsigned long long w = __hip_atomic_load((unsigned long long*)a.summary + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long c = w & ~3ull;
            const unsigned long long fin = !on ? 0ull : (w & 2ull) ? 2ull : (c | (c ? 1ull : 0ull));
            a.summary[i] = fin;
            if (a.host_summary) a.host_summary[i] = fin;
            if (FULL && a.status && on && (w & 2ull))
              for (int t = 0; t < a.T; ++t) a.status[(uint64_t)i * (uint32_t)a.T + t] = 255;
          }
          if (lane == 0) a.ticket[wt] = 0u;  // ready for the next launch
        }
      } else if (in) {
        const unsigned long long c = my + cnt[lane];
        if (last) {
          a.summary[si] = !on ? 0ull : pod_err ? 2ull : (c | (c ? 1ull : 0ull));
          if (FULL && a.status && pod_err)
            for (int t = 0; t < a.T; ++t) a.status[(uint64_t)si * (uint32_t)a.T + t] = 255;
        } else {
          *carry_w = c | (pod_err ? 2ull : 0ull);
        }
      }
#ifdef KT_PROFILE_PHASES
      __builtin_amdgcn_s_waitcnt(0);
      KT_PROF_T(t_d1);
      KT_PROF_ADD(5, t_d1 - t_s1);
#endif
    }
    if constexpr (AGG) {  // this workgroup's table as its slab (coalesced 16-byte stores), stamped with the launch's epoch
      __syncthreads();
      const uint32_t tab_bytes = (ch.n_thr * a.pk_rec + 15u) & ~15u;
      u32x4* dst = (u32x4*)(a.slab + (size_t)ch.slab_off * 16 + (size_t)blockIdx.x * tab_bytes);
      lds_u4p src = (lds_u4p)(lds + a.off_tab);
      for (uint32_t i = threadIdx.x; i < tab_bytes / 16; i += kBlockIx) dst[i] = src[i];
      if (threadIdx.x == 0) a.slab_tag[ci * kSlabTagStride + blockIdx.x] = a.epoch;
    }
  }
#ifdef KT_PROFILE_PHASES
  if (!SMALL && lane == 0) {
    KT_PROF_T(t_kernel1);
    for (int k = 0; k < 8; ++k) atomicAdd(&kt_prof_check[k], prof_acc[k]);
    atomicAdd(&kt_prof_check[8], t_kernel1 - t_kernel0);
    atomicAdd(&kt_prof_check[9], 1ull);
  }
#endif
}

#define KT_BM_LAUNCH(DT_, LA_, VETO_, NEED_, WPE_, FULL_, SMALL_, ONE_)                                         \
  {                                                                                                             \
    auto kfn = kt_check_bitmap<DT_, LA_, VETO_, NEED_, WPE_, FULL_, SMALL_, ONE_>;                              \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);    \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bytes, s, bm_args);                                                     \
  }
#define KT_BM_CASE(DT_, LA_, VETO_, NEED_)                                                   \
  {                                                                                          \
    if (small) KT_BM_LAUNCH(DT_, LA_, VETO_, NEED_, 4, true, true, false)                    \
    else if (full) KT_BM_LAUNCH(DT_, LA_, VETO_, NEED_, 4, true, false, false)               \
    else if (two_per_cu) KT_BM_LAUNCH(DT_, LA_, VETO_, NEED_, 8, false, false, true)         \
    else KT_BM_LAUNCH(DT_, LA_, VETO_, NEED_, 4, false, false, false)                        \
  }

// programs whose terms count four or five positive keys (ix.max_need > 3): the NEED = 5 instantiations — small, full and the
// lean one-workgroup-per-CU form (the two-per-CU and fused-sweep forms are not built for them)
#define KT_BM_CASE5(DT_, LA_)                                                           \
  {                                                                                     \
    if (small) KT_BM_LAUNCH(DT_, LA_, true, 5, 4, true, true, false)                    \
    else if (full) KT_BM_LAUNCH(DT_, LA_, true, 5, 4, true, false, false)               \
    else KT_BM_LAUNCH(DT_, LA_, true, 5, 4, false, false, false)                        \
  }

// returns the dispatched kernel's symbol, or nullptr when a chunk of the index does not fit the workgroup's LDS
// beside the working buffers (a single throttle with thousands of terms)
const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s, const CheckSmall* sm, bool overflow_pods, const CheckByNs* by_ns, bool one_per_cu) {
  if (n <= 0) return "";
  const int DT = dt_bucket_ix(pods.D), LA = pods.LA;
  if (status) (void)hipMemsetAsync(status, 0, (size_t)n * (size_t)sp.T, s);
  uint32_t bm_total = 0;
  BmCheckArgs bm_args = make_bm_check_args(pods, n, rows_dev, sp, sp_dev, ix, recs, summary, status, &bm_total);
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  const size_t lds_bytes = bm_total;
  const bool small = sm != nullptr && n <= kCheckSmallMax;
  if (small) {
    (void)hipMemsetAsync(summary, 0, (size_t)n * 8, s);  // the counters meet by atomics
    bm_args.ticket = sm->ticket, bm_args.host_summary = sm->host_summary;
    bm_args.n_inline = sm->n_inline;
    for (int k = 0; k < 8; ++k) bm_args.inline_rows[k] = sm->inline_rows[k];
  }
  bm_args.has_overflow = overflow_pods ? 1u : 0u;
  if (by_ns && !small && rows_dev) {
    bm_args.ix.by_ns = 1u;
    bm_args.v_meta = by_ns->v_meta, bm_args.v_latom = by_ns->v_latom, bm_args.carry = by_ns->carry;
    bm_args.wv_img = by_ns->wv_img, bm_args.wv_total_words = by_ns->wv_total_words;
    bm_args.wg_range = by_ns->wg_range_G == check_sweep_blocks(n) ? by_ns->wg_range : nullptr;
  }
  // the lean instantiation serves the PreFilter sweep; a throttle whose run of term numbers spans words (ix.has_long) needs the
  // match-by-match peel of the full one ("reported once" across words)
  const bool full = status != nullptr || ix.n_slow != 0 || overflow_pods || ix.has_long;
  const bool need5 = ix.max_need > 3u;
  // Two workgroups per CU (8 waves per SIMD, 64 VGPRs) for the ONE form — the index is one chunk scanned in row order and two
  // LDS footprints fit — unless the caller asks for one (KT_CHECK_ONE_PER_CU: A/B runs and a parity test).  A multi-chunk sweep
  // always runs one per CU: the generic 64-VGPR form carried 116-330 B of scratch, and where its LDS would have allowed it (a
  // 16-dimension engine whose plain fold cuts the chunks small) one per CU is the faster one — 76.3 -> 67.4 us at 1M x 1k, D = 16
  // (round 6; the instantiation is gone).
  const bool one = ix.n_chunks == 1 && bm_args.ix.by_ns == 0u;
  // (16 / 32 atom slots: the 64-VGPR form carries 92-330 B of scratch where the 128-VGPR one has 20-144, at the same time —
  //  0.0805 against 0.0803 ms at 1M x 1k with 16 labels per pod: one per CU there)
  const bool two_per_cu = one && !full && !small && !need5 && !one_per_cu && LA <= 8 && 2 * bm_total <= (uint32_t)kMaxLds;
  int64_t nb = (n + kBlockIx - 1) / kBlockIx;
  const int64_t max_b = two_per_cu ? 2 * kCUs : kCUs;
  if (nb > max_b) nb = max_b;
  if (nb != check_sweep_blocks(n)) bm_args.wg_range = nullptr;  // (two workgroups per CU: the ranges were planned for one)
  dim3 g_((unsigned)nb), b_(kBlockIx);
  if (small) g_ = dim3(ix.n_chunks, (unsigned)((n + kWave - 1) / kWave));
  static const bool dbg_lds = getenv("KT_DEBUG_LDS") != nullptr;
  if (dbg_lds) fprintf(stderr, "kt_check_bitmap: lds=%u (%d per CU) chunks=%u LA=%d veto=%u need=%u\n", bm_total, two_per_cu ? 2 : 1, ix.n_chunks, LA, ix.has_veto, ix.max_need);
  const bool rich = ix.rich;
#ifdef KT_FAST_BUILD
  KT_BM_CASE(8, 8, false, 2)
#else
  if (!rich) { if (DT <= 8) KT_BM_CASE(8, 8, false, 2) else KT_BM_CASE(16, 8, false, 2) }
  else if (need5) {
    if (LA <= 8) { if (DT <= 8) KT_BM_CASE5(8, 8) else KT_BM_CASE5(16, 8) }
    else if (LA <= 16) { if (DT <= 8) KT_BM_CASE5(8, 16) else KT_BM_CASE5(16, 16) }
    else { if (DT <= 8) KT_BM_CASE5(8, 32) else KT_BM_CASE5(16, 32) }
  }
  else if (LA <= 8) { if (DT <= 8) KT_BM_CASE(8, 8, true, 3) else KT_BM_CASE(16, 8, true, 3) }
  else if (LA <= 16) { if (DT <= 8) KT_BM_CASE(8, 16, true, 3) else KT_BM_CASE(16, 16, true, 3) }
  else { if (DT <= 8) KT_BM_CASE(8, 32, true, 3) else KT_BM_CASE(16, 32, true, 3) }
#endif
  return small ? "kt_check_bitmap_small" : ix.n_chunks == 1 ? "kt_check_bitmap" : "kt_check_bitmap_chunked";
}

// kt_sweep: PreFilter sweep of rows [0, n) + the packed reconcile scan of the same rows in one launch (AGG instantiation).
// nullptr: not dispatchable (several chunks, a slow list, LDS) — the caller runs check and aggregate one after the other.
#define KT_SWEEP_CASE(DT_, LA_, VETO_, NEED_)                                                                   \
  {                                                                                                             \
    auto kfn = kt_check_bitmap<DT_, LA_, VETO_, NEED_, 4, false, false, true, true>;                            \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);    \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bytes, s, bm_args);                                                     \
  }
const char* launch_sweep_indexed(const PodTable& pods, int64_t n, const SelProgram& sp, const SelProgram* sp_dev, const IndexDev& ix,
                                 const void* recs, uint64_t* summary, const PackPlan& pk, void* slab, uint32_t* slab_tag, uint32_t epoch,
                                 int* launched_blocks, hipStream_t s) {
  if (n <= 0 || ix.n_chunks != 1 || ix.n_slow != 0 || ix.has_long || ix.max_need > 3u || pk.nw == 0) return nullptr;
  const int DT = dt_bucket_ix(pods.D), LA = pods.LA;
  if (DT != 8) return nullptr;  // (the packed fold's records are planned for the 8-dimension instantiation, as in kt_aggregate_bitmap)
  uint32_t bm_total = 0;
  BmCheckArgs bm_args = make_bm_check_args(pods, n, nullptr, sp, sp_dev, ix, recs, summary, nullptr, &bm_total, &pk);
  if (bm_total > (uint32_t)kMaxLds) return nullptr;
  bm_args.slab = (unsigned char*)slab, bm_args.slab_tag = slab_tag, bm_args.epoch = epoch;
  const size_t lds_bytes = bm_total;
  // (the grid the caller sized the PackPlan's fields for: one slab per workgroup, at most kSlabTagStride of them)
  const int64_t nb = aggregate_blocks(n);
  dim3 g_((unsigned)nb), b_(kBlockIx);
  if (!ix.rich) KT_SWEEP_CASE(8, 8, false, 2)
  else if (LA <= 8) KT_SWEEP_CASE(8, 8, true, 3)
  else if (LA <= 16) KT_SWEEP_CASE(8, 16, true, 3)
  else KT_SWEEP_CASE(8, 32, true, 3)
  *launched_blocks = (int)nb;
  return "kt_sweep_bitmap";
}

}  // namespace kt
