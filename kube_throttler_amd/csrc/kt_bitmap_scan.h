// kt_bitmap_scan.h — the wave-autonomous selector scan shared by kt_check_bitmap and kt_aggregate_bitmap.
//
// The bitmap form of the selector program (kt_index.h) lives in LDS.  One wave owns a tile of 64 pods
// (lane = pod) and produces the tile's (pod, throttle) matches as a dense list in its private LDS area:
//
//   advance : every lane that still has words takes the next 64-bit word w its namespace can touch and forms
//                 x = (rows[0] | OR_l rows[row(label_l)])[w] & nsrows[ns][w]          (candidate terms)
//   peel    : while any lane holds candidate bits, each such lane takes its lowest bit, reads the
//             16-byte TermRec and decides the term; the matches of the step are appended to the list
//             with one ballot + mbcnt (no atomics; the list length lives in a scalar register)
//   drain   : whenever the list could overflow on the next step, and once at the end, the caller's
//             consumer runs over the list with its own lane mapping.
//
// All control flow is wave-uniform (ballots and scalar counters); per-lane work is predicated, not
// branched, and every load is issued from an always-valid address so that nothing needs an exec-mask
// region.  The rare term shapes (extra requirements, multi-term throttles, unconvertible selectors) are
// the only divergent code and cost nothing when absent.
#pragma once
#include "kt_index_device.h"

namespace kt {


// LDS staging plan of the bitmap tables: the index keeps them as one blob (IndexDev::bm_blob, the LDS image)
struct BmIndexArgs {
  const unsigned char* blob;
  uint32_t blob_bytes;  // multiple of 16
  uint32_t lds_off;     // where the blob goes in LDS
  uint32_t off[6];      // rows, nsrows, nswords_off, nswords, buckets, trec — relative to lds_off
  uint32_t stride, bucket_mask;
};

template <class Take>
static inline void plan_bitmap_index(const IndexDev& ix, BmIndexArgs& a, Take&& take) {
  a.blob = ix.bm_blob, a.blob_bytes = ix.bm_blob_bytes;
  a.lds_off = take(ix.bm_blob_bytes);
  for (int k = 0; k < 6; ++k) a.off[k] = ix.bm_off[k];
  a.stride = ix.bm_stride, a.bucket_mask = ix.bm_bucket_mask;
}

typedef KT_LDS const uint64_t* lds_u64p;
struct BmView {
  lds_u64p rows, nsrows;  // 64-bit bitmap words
  lds_u32p nsw_off, nsw;
  lds_u4p buckets, trec;
  uint32_t stride, bucket_mask;
};

// copies the blob into LDS (all threads of the workgroup; the caller barriers afterwards): four independent
// 16-byte loads per thread are in flight before the first store, so the whole image costs about one memory
// round trip per 64 KB
__device__ __forceinline__ BmView stage_bitmap_index(KT_LDS unsigned char* lds, const BmIndexArgs& a) {
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
  BmView v;
  v.rows = (lds_u64p)(base + a.off[0]);
  v.nsrows = (lds_u64p)(base + a.off[1]);
  v.nsw_off = (lds_u32p)(base + a.off[2]);
  v.nsw = (lds_u32p)(base + a.off[3]);
  v.buckets = (lds_u4p)(base + a.off[4]);
  v.trec = (lds_u4p)(base + a.off[5]);
  v.stride = a.stride, v.bucket_mask = a.bucket_mask;
  return v;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One 64-pod tile.  `ns` must be a valid namespace row for EVERY lane (callers pass 0 for lanes without a pod).
//   lane_match : the lane's pod takes part in selector matching (its matches are listed)
//   lane_slow  : the lane's pod is walked against throttles with an unconvertible podSelector term
//   drain(n)   : consume list[0..n)   entries = lane << 20 | throttle row
//   slow_err(t): the walk of slow throttle t hit the bad term before a match (lane-divergent call)
//   CAP        : capacity of `list` (entries); a step appends at most 64, the list is drained above CAP - 64
template <int LT, bool KEYS, uint32_t CAP, class Drain, class SlowErr>
__device__ __forceinline__ void bitmap_scan_tile(const BmView& b, const SelProgram* sp_dev, const uint32_t* slow_thr,
                                                 uint32_t n_slow, bool lane_match, bool lane_slow, uint32_t ns,
                                                 const uint32_t (&lp)[LT], const uint32_t (&lk)[LT], lds_u32wp list,
                                                 uint32_t lane, Drain&& drain, SlowErr&& slow_err) {
  uint32_t rp[LT], rk[LT];  // word offsets of the label rows
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    rp[l] = atom_row(b.buckets, b.bucket_mask, lp[l]) * b.stride;
    rk[l] = KEYS ? atom_row(b.buckets, b.bucket_mask, lk[l] ? (kKeyAtom | lk[l]) : 0u) * b.stride : b.stride;
  }
  uint32_t k = b.nsw_off[ns];
  const uint32_t k1 = lane_match ? b.nsw_off[ns + 1] : k;
  const uint32_t nsbase = ns * b.stride;
  uint64_t x = 0;
  uint32_t w = 0;
  uint32_t ks = 0, n_list = 0;  // wave-uniform
  bool more = true;
  // decides candidate term c for this lane's pod (c = 0 with has = false for idle lanes)
  auto decide = [&](bool has, uint32_t c, uint32_t& t) -> bool {
    const u32x4 tr = b.trec[c];  // {g, t, pair2, flags}
    bool hasp = false;
#pragma unroll
    for (int l = 0; l < LT; ++l) hasp |= lp[l] == tr.z;
    bool ok = has && (!(tr.w & kPostPair2) || hasp);
    if (ok && (tr.w & (kPostComplex | kPostMulti))) {  // rare shapes: generic requirement walk
      const SelProgram& sp = *sp_dev;
      const Matcher<LT, KEYS> m{sp, sp.ns_term_ok + (size_t)ns * sp.gw, lp, lk};
      ok = m.rare(tr.x, tr.y, tr.w);
    }
    t = tr.y;
    return ok;
  };
  do {
    while (n_list <= CAP - kWave) {
      if (__ballot(x != 0) != 0ull) {
        // ---- peel: one candidate term per lane that has any
        const bool has = x != 0;
        const uint32_t c = has ? w * 64u + (uint32_t)__ffsll((unsigned long long)x) - 1u : 0u;
        x &= x - 1ull;
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
        // ---- advance: next word of every lane that still has one
        const bool adv = k < k1;
        w = b.nsw[adv ? k : 0u];
        uint64_t xx = b.rows[w];  // row 0: terms without a positive requirement
#pragma unroll
        for (int l = 0; l < LT; ++l) {
          xx |= b.rows[rp[l] + w];
          if (KEYS) xx |= b.rows[rk[l] + w];
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
