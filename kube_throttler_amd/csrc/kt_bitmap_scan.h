// kt_bitmap_scan.h — the wave-autonomous selector scan shared by kt_check_bitmap and kt_aggregate_bitmap.
//
// The bitmap form of the selector program (kt_index.h) is cut into chunks that fit LDS; the kernels make one chunk
// resident at a time (open_chunk) and scan every tile of the workgroup against it.  One wave owns a tile of 64 pods
// (lane = pod) and produces the tile's (pod, throttle) matches as a dense list in its private LDS area:
//
//   advance : every lane that still has words takes the next 64-bit word w its namespace can touch in this chunk
//             and forms
//                 x = (rows[0] | OR_l rows[row(label_l)])[w] & nsrows[ns][w]          (candidate terms)
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

// What the kernels need of the index (IndexDev): the chunk directory, the images, the atom buckets
struct BmIndexArgs {
  const unsigned char* blob;   // chunk images
  const BmChunk* chunks;
  const u32x4* buckets;        // AtomBucket table (staged once)
  uint32_t n_chunks, bucket_bytes;
  uint32_t lds_buckets, lds_img;  // LDS offsets: buckets, current chunk image
  uint32_t bucket_mask, bucket_mult, key_rows, has_inline;
};

template <class Take>
static inline void plan_bitmap_index(const IndexDev& ix, BmIndexArgs& a, Take&& take) {
  a.blob = ix.bm_blob, a.chunks = ix.bm_chunks, a.buckets = (const u32x4*)ix.bm_buckets;
  a.n_chunks = ix.n_chunks, a.bucket_bytes = ix.bm_bucket_bytes;
  a.lds_buckets = take(ix.bm_bucket_bytes);
  a.lds_img = take(ix.bm_max_img);
  a.bucket_mask = ix.bm_bucket_mask, a.bucket_mult = ix.bm_bucket_mult, a.key_rows = ix.bm_has_key_rows;
  a.has_inline = ix.bm_has_inline;
}

// The tables of the chunk that is resident in LDS
struct BmView {
  KT_LDS const unsigned long long* rows;    // [rows][stride] 64-bit bitmap words of this chunk
  KT_LDS const unsigned long long* nsrows;  // [n_ns][stride]
  lds_u32p nsb_off, nsb;                    // chunk-local word indices a namespace can touch
  lds_u4p buckets, trec, trecx;
  uint32_t stride, bucket_mask, bucket_mult, key_rows, has_inline;
};

// all threads of the workgroup: n16 16-byte pieces from src to dst, four independent loads in flight per thread
__device__ __forceinline__ void lds_stage16(KT_LDS u32x4* dst, const u32x4* src, uint32_t n16) {
  for (uint32_t i = threadIdx.x; i < n16; i += 4 * kBlockIx) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[min(i + k * kBlockIx, n16 - 1u)];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * kBlockIx < n16) dst[i + k * kBlockIx] = v[k];
  }
}

// Makes chunk `ch` resident (the caller barriers before — nobody still reads the previous image — and after)
__device__ __forceinline__ BmView open_chunk(KT_LDS unsigned char* lds, const BmIndexArgs& a, const BmChunk& ch) {
  lds_stage16((KT_LDS u32x4*)(lds + a.lds_img), (const u32x4*)(a.blob + ch.img_off), ch.img_bytes / 16u);
  KT_LDS unsigned char* base = lds + a.lds_img;
  BmView v;
  v.rows = (KT_LDS const unsigned long long*)base;
  v.nsrows = (KT_LDS const unsigned long long*)(base + ch.off_nsrows);
  v.nsb_off = (lds_u32p)(base + ch.off_nsw_off);
  v.nsb = (lds_u32p)(base + ch.off_nsw);
  v.trec = (lds_u4p)(base + ch.off_trec);
  v.trecx = (lds_u4p)(base + ch.off_trecx);
  v.buckets = (lds_u4p)(lds + a.lds_buckets);
  v.stride = ch.stride, v.bucket_mask = a.bucket_mask, v.bucket_mult = a.bucket_mult, v.key_rows = a.key_rows;
  v.has_inline = a.has_inline;
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

// One 64-pod tile.  `ns` must be a valid namespace row for EVERY lane (callers pass 0 for lanes without a pod).
//   lane_match : the lane's pod takes part in selector matching (its matches are listed)
//   lane_slow  : the lane's pod is walked against throttles with an unconvertible podSelector term
//   drain(n)   : consume list[0..n)   entries = lane << 20 | throttle row (BY_RANK: chunk-local throttle rank)
//   slow_match(t): BY_RANK only — the pod matches slow throttle t (which has no rank): lane-divergent call
//   slow_err(t): the walk of slow throttle t hit the bad term before a match (lane-divergent call)
//   CAP        : capacity of `list` (entries); a step appends at most 64, the list is drained above CAP - 64
template <int LT, bool KEYS, uint32_t CAP, bool BY_RANK, class Drain, class SlowErr, class SlowMatch>
__device__ __forceinline__ void bitmap_scan_tile(const BmView& b, const SelProgram* sp_dev, const uint32_t* slow_thr,
                                                 uint32_t n_slow, bool lane_match, bool lane_slow, uint32_t ns,
                                                 const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], lds_u32wp list,
                                                 uint32_t lane, Drain&& drain, SlowErr&& slow_err, SlowMatch&& slow_match) {
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
  uint64_t x = 0;
  uint32_t w = 0, last_t = 0xFFFFFFFFu;
  uint32_t ks = 0, n_list = 0;  // wave-uniform
  bool more = true;
  // decides candidate term c for this lane's pod (c = 0 with has = false for idle lanes)
  auto decide = [&](bool has, uint32_t c, uint32_t& t, uint32_t& rank) -> bool {
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
      const Matcher<LT, KEYS> m{*sp_dev, lp, lk};
      ok = m.rare(tr.x);
    }
    t = tr.y;
    rank = tr.w >> 8;  // chunk-local throttle rank
    if ((tr.w & kPostAdj) && t == last_t) ok = false;  // an earlier term of the same throttle matched already
    if (ok) last_t = t;
    return ok;
  };
  do {
    while (n_list <= CAP - kWave) {
      const bool has = x != 0;
      if (__ballot(has) != 0ull) {
        // ---- peel: one candidate term per lane that has any
        const uint32_t c = has ? w * 64u + (uint32_t)__ffsll((unsigned long long)x) - 1u : 0u;
        x &= x - 1ull;
        uint32_t t = 0, rank = 0;
        const bool ok = decide(has, c, t, rank);
        const uint64_t mk = __ballot(ok);
        if (ok) list[n_list + lane_rank(mk)] = lane << 20 | (BY_RANK ? rank : t);
        n_list += (uint32_t)__popcll(mk);
      } else if (ks < n_slow) {
        // ---- throttles with an unconvertible podSelector term: in-order walk (error semantics depend on
        //      term order, throttle_selector.go:30-42)
        const SelProgram& sp = *sp_dev;
        const int ts = (int)slow_thr[ks++];
        const uint32_t res = walk_slow<LT, KEYS>(sp, ts, sp.ns_term_ok + (size_t)ns * sp.gw, lane_slow, lp, lk);
        if (res & kSlowError) slow_err((uint32_t)ts);
        const bool ok = (res & kSlowMatched) && lane_match;
        if (BY_RANK) {
          if (ok) slow_match((uint32_t)ts);
        } else {
          const uint64_t mk = __ballot(ok);
          if (ok) list[n_list + lane_rank(mk)] = lane << 20 | (uint32_t)ts;
          n_list += (uint32_t)__popcll(mk);
        }
      } else if (__ballot(k < k1) != 0ull) {
        // ---- advance: next block of every lane that still has one
        const bool adv = k < k1;
        w = b.nsb[adv ? k : 0u];
        unsigned long long xx = b.rows[w];  // row 0: terms without a positive requirement
#pragma unroll
        for (int l = 0; l < LT; ++l) {
          xx |= b.rows[rp[l] + w];
          if (key_rows) xx |= b.rows[rk[l] + w];
        }
        xx &= b.nsrows[nsbase + w];
        x = adv ? xx : 0ull;
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
