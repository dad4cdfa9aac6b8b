// kt_bitmap_scan.h — the wave-autonomous selector scan shared by kt_check_bitmap and kt_aggregate_bitmap.
//
// The bitmap form of the selector program (kt_index.h) lives in LDS when it fits (small-T regime), else it is
// read through L2 from the index blob.  One wave owns a tile of 64 pods (lane = pod) and produces the tile's
// (pod, throttle) matches as a dense list in its private LDS area:
//
//   advance : every lane that still has steps takes the next 64-bit word (LDS form) / 128-bit block (L2 form) b its
//             namespace can touch and forms
//                 x = (rows[0] | OR_l rows[row(label_l)])[b] & nsrows[ns][b]          (candidate terms)
//   peel    : while any lane holds candidate bits, each such lane takes its lowest bit, reads the
//             16-byte TermRec and decides the term (second matchLabels pair: 8 compares; up to two more small
//             requirements from the 32-byte TermX; anything else: the generic requirement walk); the matches
//             of the step are appended to the list with one ballot + mbcnt (no atomics; the list length lives
//             in a scalar register)
//   drain   : whenever the list could overflow on the next step, and once at the end, the caller's
//             consumer runs over the list with its own lane mapping.
//
// All control flow is wave-uniform (ballots and scalar counters); per-lane work is predicated, not
// branched, and every load is issued from an always-valid address so that nothing needs an exec-mask
// region.  A throttle with several selector terms is reported once: its terms are numbered contiguously and a
// lane meets its candidates in ascending number, so a match that repeats the lane's previous throttle is dropped.
#pragma once
#include "kt_index_device.h"

namespace kt {

// Staging plan of the bitmap tables: the index keeps them as one blob (IndexDev::bm_blob, the LDS image)
struct BmIndexArgs {
  const unsigned char* blob;
  uint32_t blob_bytes;  // staged prefix (LDS form), multiple of 16
  uint32_t lds_off;     // where the blob goes in LDS (LDS-resident form)
  uint32_t off[9];      // rows, nsrows, nsblocks_off, nsblocks, buckets, trec, trecx, nswords64_off, nswords64
  uint32_t stride, words, bucket_mask, bucket_mult, key_rows, has_inline;  // stride / words: 64-bit words per row
};

template <class Take>
static inline void plan_bitmap_index(const IndexDev& ix, BmIndexArgs& a, bool in_lds, Take&& take) {
  a.blob = ix.bm_blob, a.blob_bytes = ix.bm_lds_bytes;
  a.lds_off = in_lds ? take(ix.bm_lds_bytes) : 0u;
  for (int k = 0; k < 9; ++k) a.off[k] = ix.bm_off[k];
  a.stride = ix.bm_stride, a.words = ix.bm_words, a.bucket_mask = ix.bm_bucket_mask, a.bucket_mult = ix.bm_bucket_mult, a.key_rows = ix.bm_has_key_rows;
  a.has_inline = ix.bm_has_inline;
}

typedef unsigned long long kt_u64x2 __attribute__((ext_vector_type(2)));

// The tables, in LDS (LDSIX) or in the device blob (read through L2).  The LDS form walks the bitmaps 64 bits at
// a time (LDS reads are cheap, shorter peel loops win); the L2 form 128 bits at a time (one cache-line request per
// 16 bytes: half the requests).
template <bool LDSIX>
struct BmView;
template <>
struct BmView<true> {
  typedef unsigned long long word_t;  // 64-bit steps
  static constexpr uint32_t kBits = 64;
  KT_LDS const word_t* rows;
  KT_LDS const word_t* nsrows;
  lds_u32p nsb_off, nsb;  // 64-bit word indices per namespace
  lds_u4p buckets, trec, trecx;
  uint32_t stride, words, bucket_mask, bucket_mult, key_rows, has_inline;
};
template <>
struct BmView<false> {
  typedef kt_u64x2 word_t;  // 128-bit steps: 16-byte reads at 8-byte alignment
  static constexpr uint32_t kBits = 128;
  const unsigned long long* rows;
  const unsigned long long* nsrows;
  const uint32_t* nsb_off;  // 128-bit block indices per namespace
  const uint32_t* nsb;
  const u32x4* buckets;
  const u32x4* trec;
  const u32x4* trecx;
  uint32_t stride, words, bucket_mask, bucket_mult, key_rows, has_inline;
};

// LDS-resident form: copies the blob into LDS (all threads of the workgroup; the caller barriers afterwards):
// four independent 16-byte loads per thread are in flight before the first store, so the whole image costs about
// one memory round trip per 64 KB.  L2 form: just the pointers.
template <bool LDSIX>
__device__ __forceinline__ BmView<LDSIX> open_bitmap_index(KT_LDS unsigned char* lds, const BmIndexArgs& a);
template <>
__device__ __forceinline__ BmView<true> open_bitmap_index<true>(KT_LDS unsigned char* lds, const BmIndexArgs& a) {
  const u32x4* src = (const u32x4*)a.blob;
  KT_LDS u32x4* dst = (KT_LDS u32x4*)(lds + a.lds_off);
  const uint32_t n16 = a.blob_bytes / 16u;
  for (uint32_t i = threadIdx.x; i < n16; i += 4 * kBlockIx) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[min(i + k * kBlockIx, n16 - 1u)];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * kBlockIx < n16) dst[i + k * kBlockIx] = v[k];
  }
  KT_LDS unsigned char* base = lds + a.lds_off;
  BmView<true> v;
  v.rows = (KT_LDS const unsigned long long*)(base + a.off[0]);
  v.nsrows = (KT_LDS const unsigned long long*)(base + a.off[1]);
  v.nsb_off = (lds_u32p)(base + a.off[7]);
  v.nsb = (lds_u32p)(base + a.off[8]);
  v.buckets = (lds_u4p)(base + a.off[4]);
  v.trec = (lds_u4p)(base + a.off[5]);
  v.trecx = (lds_u4p)(base + a.off[6]);
  v.stride = a.stride, v.words = a.words, v.bucket_mask = a.bucket_mask, v.bucket_mult = a.bucket_mult, v.key_rows = a.key_rows, v.has_inline = a.has_inline;
  return v;
}
template <>
__device__ __forceinline__ BmView<false> open_bitmap_index<false>(KT_LDS unsigned char*, const BmIndexArgs& a) {
  const unsigned char* base = a.blob;
  BmView<false> v;
  v.rows = (const unsigned long long*)(base + a.off[0]);
  v.nsrows = (const unsigned long long*)(base + a.off[1]);
  v.nsb_off = (const uint32_t*)(base + a.off[2]);
  v.nsb = (const uint32_t*)(base + a.off[3]);
  v.buckets = (const u32x4*)(base + a.off[4]);
  v.trec = (const u32x4*)(base + a.off[5]);
  v.trecx = (const u32x4*)(base + a.off[6]);
  v.stride = a.stride, v.words = a.words, v.bucket_mask = a.bucket_mask, v.bucket_mult = a.bucket_mult, v.key_rows = a.key_rows, v.has_inline = a.has_inline;
  return v;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// branch-free 4-way bucket probe: bitmap row of `atom`, or row 1 (all zero) when no selector is anchored on it
template <class P4>
__device__ __forceinline__ uint32_t atom_row_of(P4 buckets, uint32_t mask, uint32_t mult, uint32_t atom) {
  const uint32_t b = atom_bucket(atom, mask, mult);
  const u32x4 a = buckets[2 * b], r = buckets[2 * b + 1];
  uint32_t row = 1u;
  row = a.x == atom ? r.x : row;
  row = a.y == atom ? r.y : row;
  row = a.z == atom ? r.z : row;
  row = a.w == atom ? r.w : row;
  return atom ? row : 1u;
}

// one inline extra requirement (TermX::e[k]) against the pod's label registers
template <int LT, bool KEYS>
__device__ __forceinline__ bool extra_ok(const u32x4 e, const uint32_t (&lp)[LT], const uint32_t (&lk)[LT]) {
  const uint32_t op = e.x & 0xFFu;
  const bool key_op = op >= kOpExists;
  bool hit = false;
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const uint32_t v = (KEYS && key_op) ? lk[l] : lp[l];
    hit |= (v == e.y) | (v == e.z) | (v == e.w);
  }
  const bool positive = op == kOpIn || op == kOpExists;
  return op == 0xFFu || (positive ? hit : !hit);
}

// one step of a bitmap row: word w (LDS form) / the 16 bytes of block b = words 2b, 2b+1 (L2 form)
__device__ __forceinline__ unsigned long long load_step(const BmView<true>& b, KT_LDS const unsigned long long* row, uint32_t step) {
  return row[step];
}
__device__ __forceinline__ kt_u64x2 load_step(const BmView<false>& b, const unsigned long long* row, uint32_t step) {
  return *(const kt_u64x2*)(row + 2 * step);
}
// candidate bits of one step as a (lo, hi) pair of 64-bit halves
__device__ __forceinline__ void split_step(const BmView<true>&, unsigned long long v, uint32_t, uint64_t& lo, uint64_t& hi) { lo = v, hi = 0; }
__device__ __forceinline__ void split_step(const BmView<false>& b, kt_u64x2 v, uint32_t step, uint64_t& lo, uint64_t& hi) {
  lo = v.x, hi = 2 * step + 1 < b.words ? v.y : 0ull;  // an odd word count leaves the last block half empty
}

// One 64-pod tile.  `ns` must be a valid namespace row for EVERY lane (callers pass 0 for lanes without a pod).
//   lane_match : the lane's pod takes part in selector matching (its matches are listed)
//   lane_slow  : the lane's pod is walked against throttles with an unconvertible podSelector term
//   drain(n)   : consume list[0..n)   entries = lane << 20 | throttle row
//   slow_err(t): the walk of slow throttle t hit the bad term before a match (lane-divergent call)
//   CAP        : capacity of `list` (entries); a step appends at most 64, the list is drained above CAP - 64
template <int LT, bool KEYS, uint32_t CAP, class View, class Drain, class SlowErr>
__device__ __forceinline__ void bitmap_scan_tile(const View& b, const SelProgram* sp_dev, const uint32_t* slow_thr,
                                                 uint32_t n_slow, bool lane_match, bool lane_slow, uint32_t ns,
                                                 const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], lds_u32wp list,
                                                 uint32_t lane, Drain&& drain, SlowErr&& slow_err) {
  uint32_t rp[LT], rk[LT];  // word offsets of the label rows
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    rp[l] = atom_row_of(b.buckets, b.bucket_mask, b.bucket_mult, lp[l]) * b.stride;
    rk[l] = KEYS ? atom_row_of(b.buckets, b.bucket_mask, b.bucket_mult, lk[l] ? (kKeyAtom | lk[l]) : 0u) * b.stride : b.stride;
  }
  const bool key_rows = KEYS && b.key_rows != 0;  // some term is anchored on Exists{key}
  uint32_t k = b.nsb_off[ns];
  const uint32_t k1 = lane_match ? b.nsb_off[ns + 1] : k;
  const uint32_t nsbase = ns * b.stride;
  uint64_t xlo = 0, xhi = 0;
  uint32_t blk = 0, last_t = 0xFFFFFFFFu;
  uint32_t ks = 0, n_list = 0;  // wave-uniform
  bool more = true;
  // decides candidate term c for this lane's pod (c = 0 with has = false for idle lanes)
  auto decide = [&](bool has, uint32_t c, uint32_t& t) -> bool {
    const u32x4 tr = b.trec[c];  // {g, t, pair2, flags}
    bool hasp = false;
#pragma unroll
    for (int l = 0; l < LT; ++l) hasp |= lp[l] == tr.z;
    bool ok = has && (!(tr.w & kPostPair2) || hasp);
    if (b.has_inline && __ballot(ok && (tr.w & kPostInline)) != 0ull) {  // wave-uniform: free when no such term exists
      const u32x4 e0 = b.trecx[2 * c], e1 = b.trecx[2 * c + 1];
      const bool eok = extra_ok<LT, KEYS>(e0, lp, lk) && extra_ok<LT, KEYS>(e1, lp, lk);
      ok = ok && (!(tr.w & kPostInline) || eok);
    }
    if (ok && (tr.w & kPostComplex)) {  // rare shapes: generic requirement walk
      const SelProgram& sp = *sp_dev;
      const Matcher<LT, KEYS> m{sp, sp.ns_term_ok + (size_t)ns * sp.gw, lp, lk};
      ok = m.rare(tr.x, tr.y, tr.w);
    }
    t = tr.y;
    if ((tr.w & kPostAdj) && t == last_t) ok = false;  // an earlier term of the same throttle matched already
    if (ok) last_t = t;
    return ok;
  };
  do {
    while (n_list <= CAP - kWave) {
      const bool has = View::kBits == 64 ? xlo != 0 : (xlo | xhi) != 0;
      if (__ballot(has) != 0ull) {
        // ---- peel: one candidate term per lane that has any
        uint32_t c;
        if (View::kBits == 64) {
          c = has ? blk * 64u + (uint32_t)__ffsll((unsigned long long)xlo) - 1u : 0u;
          xlo &= xlo - 1ull;
        } else {
          const bool lo = xlo != 0;
          uint64_t v = lo ? xlo : xhi;
          c = has ? blk * 128u + (lo ? 0u : 64u) + (uint32_t)__ffsll((unsigned long long)v) - 1u : 0u;
          v &= v - 1ull;
          xlo = lo ? v : xlo;
          xhi = lo ? xhi : v;
        }
        uint32_t t = 0;
        const bool ok = decide(has, c, t);
        const uint64_t mk = __ballot(ok);
        if (ok) list[n_list + lane_rank(mk)] = lane << 20 | t;
        n_list += (uint32_t)__popcll(mk);
      } else if (ks < n_slow) {
        // ---- throttles with an unconvertible podSelector term: in-order walk (error semantics depend on
        //      term order, throttle_selector.go:30-42)
        const SelProgram& sp = *sp_dev;
        const int ts = (int)slow_thr[ks++];
        bool matched, err;
        walk_slow<LT, KEYS>(sp, ts, sp.ns_term_ok + (size_t)ns * sp.gw, lane_slow, lp, lk, matched, err);
        if (err) slow_err((uint32_t)ts);
        const bool ok = matched && lane_match;
        const uint64_t mk = __ballot(ok);
        if (ok) list[n_list + lane_rank(mk)] = lane << 20 | (uint32_t)ts;
        n_list += (uint32_t)__popcll(mk);
      } else if (__ballot(k < k1) != 0ull) {
        // ---- advance: next block of every lane that still has one
        const bool adv = k < k1;
        blk = b.nsb[adv ? k : 0u];
        typename View::word_t xx = load_step(b, b.rows, blk);  // row 0: terms without a positive requirement
#pragma unroll
        for (int l = 0; l < LT; ++l) {
          xx |= load_step(b, b.rows + rp[l], blk);
          if (key_rows) xx |= load_step(b, b.rows + rk[l], blk);
        }
        xx &= load_step(b, b.nsrows + nsbase, blk);
        uint64_t nlo, nhi;
        split_step(b, xx, blk, nlo, nhi);
        xlo = adv ? nlo : 0ull;
        xhi = adv ? nhi : 0ull;
        k += adv ? 1u : 0u;
      } else {
        more = false;
        break;
      }
    }
    if (n_list != 0) drain(n_list);
    n_list = 0;
  } while (more);
}

}  // namespace kt
