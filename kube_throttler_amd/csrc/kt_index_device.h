// kt_index_device.h — device side of the indexed pod x throttle scans for gfx950: work ~ (pods + candidate terms).
//
// lane = pod.  Each lane probes the label-atom hash index (kt_index.h) with its own labels; postings carry
// an inline description of the common term shapes, so a candidate is usually decided from one 32-byte
// record.  Decisions still cover the full P x T matrix: every pair not enumerated is "not affected" by
// construction of the index.
//
//  * kt_check_indexed     : hash slots + postings staged in LDS when they fit (one 1024-thread workgroup per
//                           CU shares one copy; 160 KB LDS/CU), else read through L2.
//  * kt_aggregate_indexed : per-workgroup partial-`used` table in LDS (ds_add_u64 / ds_add_u32), spilled to a
//                           slab and summed by kt_reduce_partials; global atomics only when the table does
//                           not fit LDS.
#pragma once
#include <cstdlib>

#include "kt_index.h"
#include "kt_kernels_common.h"
#include "kt_launch.h"

namespace kt {

constexpr int kBlockIx = 1024;       // one workgroup per CU: 16 waves = 4 per SIMD
constexpr int kMaxLds = 160 * 1024;  // gfx950 LDS per CU / per workgroup
constexpr int kCUs = 256;
constexpr uint32_t kQueueCap = 6 * 1024;  // tile match-queue entries (24 KB): ~6 matches per pod of a 1024-pod tile

template <int LT, bool KEYS>
struct Matcher {
  const SelProgram& sp;
  const uint32_t* ns_row;
  const uint32_t (&lp)[LT];
  const uint32_t (&lk)[LT];

  __device__ __forceinline__ bool ns_ok(uint32_t g) const { return (ns_row[g >> 5] >> (g & 31)) & 1u; }

  // no earlier term of throttle t matches this pod (so a throttle whose selector has several matching
  // terms is reported once — by its first matching term)
  __device__ __forceinline__ bool first_of_throttle(uint32_t g, uint32_t t) const {
    for (uint32_t g2 = sp.thr_term_off[t]; g2 < g; ++g2)
      if (ns_ok(g2) && term_match<LT, KEYS>(sp, g2, lp, lk)) return false;
    return true;
  }

  // generic path (universal lists): full requirement walk
  __device__ __forceinline__ bool owns_match(uint32_t g, bool check_ns, uint32_t& t_out) const {
    if (check_ns && !ns_ok(g)) return false;
    if (!term_match<LT, KEYS>(sp, g, lp, lk)) return false;
    const uint32_t t = sp.term_thr[g];
    if (!first_of_throttle(g, t)) return false;
    t_out = t;
    return true;
  }

  // posting fast path: the anchor requirement is already satisfied (that is how the posting was reached)
  __device__ __forceinline__ bool posting_match(const Posting& p, uint32_t ns) const {
    const uint32_t f = p.flags;
    bool ok = true;
    if (f & kPostNsMask) ok = (p.nsmask >> ns) & 1ull;
    else if (f & kPostNsBitmap) ok = ns_ok(p.g);
    if (ok && (f & kPostPair2)) {
      bool has = false;
#pragma unroll
      for (int l = 0; l < LT; ++l) has |= lp[l] == p.pair2;
      ok = has;
    }
    if (ok && (f & (kPostComplex | kPostMulti))) ok = rare(p.g, p.t, f);
    return ok;
  }

  // rare term shapes (requirements beyond one extra matchLabels pair, multi-term throttles): a real call
  __device__ __forceinline__ bool rare(uint32_t g, uint32_t t, uint32_t f) const {
    if ((f & kPostComplex) && !term_match<LT, KEYS>(sp, g, lp, lk)) return false;
    if ((f & kPostMulti) && !first_of_throttle(g, t)) return false;
    return true;
  }
};

// Explicit LDS (address space 3) pointer types: tables staged in LDS must be read with ds_read, not
// through generic/flat addressing (which costs 64-bit address math and the flat-memory latency).
#define KT_LDS __attribute__((address_space(3)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // plain vector types: loadable from any address space
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef KT_LDS const u32x4* lds_u4p;
typedef KT_LDS const u32x2* lds_u2p;
typedef KT_LDS const uint32_t* lds_u32p;
typedef KT_LDS uint32_t* lds_u32wp;
typedef KT_LDS unsigned long long* lds_u64wp;

__device__ __forceinline__ uint32_t lds_add(lds_u32wp p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add64(lds_u64wp p, unsigned long long v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// tables are viewed as arrays of 16-byte vectors: IndexSlot = 1, TermRec = 1, Posting = 2 vectors
__device__ __forceinline__ Posting make_posting(u32x4 a, u32x4 b) {
  Posting p;
  p.g = a.x, p.t = a.y, p.pair2 = a.z, p.flags = a.w;
  p.nsmask = (uint64_t)b.x | (uint64_t)b.y << 32;
  p.pad = 0;
  return p;
}
template <class V4Ptr>
__device__ __forceinline__ Posting load_posting(V4Ptr posts16, uint32_t idx) {
  return make_posting(posts16[2 * idx], posts16[2 * idx + 1]);
}

// One hash lookup: (begin, count) of the posting list filed under `key` (count 0 when absent).
template <class V4Ptr>
__device__ __forceinline__ uint2 lookup(V4Ptr slots16, uint32_t mask, uint64_t key) {
  uint32_t h = index_hash(key, mask);
  const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
  for (;;) {
    const u32x4 s = slots16[h];  // {key lo, key hi, begin, count}
    if (s.x == klo && s.y == khi) return make_uint2(s.z, s.w);
    if ((s.x | s.y) == 0) return make_uint2(0u, 0u);
    h = (h + 1) & mask;
  }
}

template <int LT, bool KEYS, class SlotPtr, class PostPtr, class F>
__device__ __forceinline__ void enumerate_matches(const SelProgram& sp, const IndexDev& ix, SlotPtr slots,
                                                  PostPtr posts, uint32_t ns, const uint32_t (&lp)[LT],
                                                  const uint32_t (&lk)[LT], F&& on_match) {
  const Matcher<LT, KEYS> m{sp, sp.ns_term_ok + (size_t)ns * sp.gw, lp, lk};
  const uint64_t scope = (uint64_t)(ns + 1) << 32;
  // all lookups first (independent loads in flight together), then the posting walks
  uint2 rn[LT], rc[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const uint32_t pair = lp[l];
    rn[l] = pair ? lookup(slots, ix.mask, scope | pair) : make_uint2(0u, 0u);    // Throttles of the pod's namespace
    rc[l] = pair ? lookup(slots, ix.mask, (uint64_t)pair) : make_uint2(0u, 0u);  // ClusterThrottles
  }
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    for (uint32_t k = 0; k < rn[l].y; ++k) {
      const Posting p = load_posting(posts, rn[l].x + k);
      if (m.posting_match(p, ns)) on_match(p.t);
    }
    for (uint32_t k = 0; k < rc[l].y; ++k) {
      const Posting p = load_posting(posts, rc[l].x + k);
      if (m.posting_match(p, ns)) on_match(p.t);
    }
  }
  if (KEYS && ix.has_key_atoms) {
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if (lk[l] == 0) continue;
      const uint32_t ka = kKeyAtom | lk[l];
      const uint2 a = lookup(slots, ix.mask, scope | ka), b = lookup(slots, ix.mask, (uint64_t)ka);
      for (uint32_t k = 0; k < a.y; ++k) {
        const Posting p = load_posting(posts, a.x + k);
        if (m.posting_match(p, ns)) on_match(p.t);
      }
      for (uint32_t k = 0; k < b.y; ++k) {
        const Posting p = load_posting(posts, b.x + k);
        if (m.posting_match(p, ns)) on_match(p.t);
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
// before a match (same semantics as the dense kernels; t is wave-uniform).
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

extern __shared__ __attribute__((aligned(16))) unsigned char kt_smem[];

__device__ __forceinline__ void lds_stage(KT_LDS unsigned char* dst, const void* src, uint32_t bytes) {
  const u32x4* s = (const u32x4*)src;
  KT_LDS u32x4* d = (KT_LDS u32x4*)dst;
  for (uint32_t i = threadIdx.x; i < (bytes + 15u) / 16u; i += kBlockIx) d[i] = s[i];
}

// branch-free 4-way bucket probe: bitmap row of `atom`, or row 1 (all zero) when no selector mentions it
__device__ __forceinline__ uint32_t atom_row(lds_u4p buckets, uint32_t mask, uint32_t atom) {
  const uint32_t b = atom_bucket(atom, mask);
  const u32x4 a = buckets[2 * b], r = buckets[2 * b + 1];
  uint32_t row = 1u;
  row = a.x == atom ? r.x : row;
  row = a.y == atom ? r.y : row;
  row = a.z == atom ? r.z : row;
  row = a.w == atom ? r.w : row;
  return atom ? row : 1u;
}


#define KT_IX_CASE(NAME, DT_, LT_, KEYS_, FLAG_)                                                                 \
  {                                                                                                             \
    auto kfn = NAME<DT_, LT_, KEYS_, FLAG_>;                                                                    \
    if (lds_bytes > 48 * 1024)                                                                                  \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
    hipLaunchKernelGGL(kfn, g_, b_, lds_bytes, s, KT_IX_ARGS);                                                  \
  }
#ifdef KT_FAST_BUILD
#define KT_IX_DISPATCH2(NAME, DT_, LT_, KEYS_, FLAG_) do { KT_IX_CASE(NAME, 8, 8, false, FLAG_) } while (0)
#else
#define KT_IX_DISPATCH2(NAME, DT_, LT_, KEYS_, FLAG_)                                                                          \
  do {                                                                                                                         \
    if (DT_ <= 8 && LT_ == 8) { if (KEYS_) KT_IX_CASE(NAME, 8, 8, true, FLAG_) else KT_IX_CASE(NAME, 8, 8, false, FLAG_) } \
    else if (DT_ <= 8) { if (KEYS_) KT_IX_CASE(NAME, 8, 16, true, FLAG_) else KT_IX_CASE(NAME, 8, 16, false, FLAG_) }           \
    else if (LT_ == 8) { if (KEYS_) KT_IX_CASE(NAME, 16, 8, true, FLAG_) else KT_IX_CASE(NAME, 16, 8, false, FLAG_) }           \
    else { if (KEYS_) KT_IX_CASE(NAME, 16, 16, true, FLAG_) else KT_IX_CASE(NAME, 16, 16, false, FLAG_) }                       \
  } while (0)
#endif


}  // namespace kt
