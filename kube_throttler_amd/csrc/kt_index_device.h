// kt_index_device.h — device-side pieces shared by the index-driven kernels (gfx950): LDS pointer types, the generic
// requirement walk for the rare term shapes the bitmaps cannot decide, the in-order walk of throttles with
// unconvertible selectors.  The scan itself is kt_scan.h.
#pragma once
#include <cstdlib>

#include "kt_index.h"
#include "kt_kernels_common.h"
#include "kt_launch.h"

namespace kt {

constexpr int kBlockIx = 1024;       // 16 waves per workgroup; one or two workgroups per CU depending on the LDS footprint
constexpr int kMaxLds = 160 * 1024;  // gfx950 LDS per CU / per workgroup
constexpr int kCUs = 256;

// Explicit LDS (address space 3) pointer types: tables staged in LDS must be read with ds_read, not
// through generic/flat addressing (which costs 64-bit address math and the flat-memory latency).
#define KT_LDS __attribute__((address_space(3)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // plain vector types: loadable from any address space
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef KT_LDS const u32x4* lds_u4p;
typedef KT_LDS const uint32_t* lds_u32p;
typedef KT_LDS uint32_t* lds_u32wp;
typedef KT_LDS unsigned long long* lds_u64wp;

__device__ __forceinline__ uint32_t lds_add(lds_u32wp p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add64(lds_u64wp p, unsigned long long v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Generic requirement walk against a pod's RAW label rows in HBM (pair ids / key ids, LS slots, 0 = empty): the rare
// paths only — candidates flagged `slow` in the index and the terms of throttles on the slow list.
//   In: the pod carries one of the requirement's pairs; NotIn: none of them (absent key included);
//   Exists: it carries the key; DoesNotExist: it does not
// (labels.Requirement.Matches of k8s.io/apimachinery v0.26.4, restated — SURVEY.md Appendix B.)
__device__ inline bool term_match_mem(const SelProgram& sp, uint32_t g, const uint32_t* lp, const uint32_t* lk, int LS) {
  bool ok = true;
  const uint32_t r1 = sp.term_req_off[g + 1];
  for (uint32_t r = sp.term_req_off[g]; r < r1 && ok; ++r) {
    const uint8_t op = sp.req_op[r];
    bool sat;
    if (op <= kOpNotIn) {
      bool in = false;
      const uint32_t j1 = sp.req_val_off[r + 1];
      for (uint32_t j = sp.req_val_off[r]; j < j1; ++j) {
        const uint32_t v = sp.req_val[j];
        for (int l = 0; l < LS; ++l) in |= lp[l] == v;
      }
      sat = (op == kOpIn) ? in : !in;
    } else {
      bool has = false;
      const uint32_t k = sp.req_key[r];
      for (int l = 0; l < LS; ++l) has |= lk[l] == k;
      sat = (op == kOpExists) ? has : !has;
    }
    ok &= sat;
  }
  return ok;
}

// Throttles with an unconvertible podSelector term: in-order walk, error when the bad term is reached
// before a match (same semantics as the dense kernels; t is wave-uniform).  Returns bit 0 = matched, bit 1 = error.
constexpr uint32_t kSlowMatched = 1u, kSlowError = 2u;
__device__ inline uint32_t walk_slow_mem(const SelProgram& sp, int t, const uint32_t* ns_row, bool lane_on, const uint32_t* lp,
                                         const uint32_t* lk, int LS) {
  uint32_t res = 0;
  bool open = lane_on;
  const uint32_t g1 = sp.thr_term_off[t + 1];
  for (uint32_t g = sp.thr_term_off[t]; g < g1; ++g) {
    const bool applies = open && ((ns_row[g >> 5] >> (g & 31)) & 1u);
    if (sp.term_flags[g] & kTermPodSelInvalid) {
      res |= applies ? kSlowError : 0u;
      open &= !applies;
      continue;
    }
    const bool mt = applies && term_match_mem(sp, g, lp, lk, LS);
    res |= mt ? kSlowMatched : 0u;
    open &= !mt;
  }
  return res;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
    v += (unsigned long long)lo | (unsigned long long)hi << 32;
  }
  return v;
}

// One record of the packed slabs (PackPlan) summed over the workgroups' slabs by ONE wave, lane = slab: every lane takes
// the record of its slabs apart into fields (the pod count, one per dimension) in 64-bit accumulators — four slabs per
// lane and trip with all their loads in flight — and the wave adds the accumulators up across its lanes.  On return every
// lane holds the totals: pods, acc[d] in field units (value = acc[d] << pk.shift[d]), zero_keys = OR of the key masks of
// pods that carry a key with the value 0.  check_tags = 0: every workgroup spilled this chunk (single-chunk programs).
struct PackedSums {
  unsigned long long acc[16], pods;
  uint32_t zero_keys;
};
__device__ __forceinline__ void packed_record_sums(const unsigned char* base, size_t pitch, int n_slabs, const PackPlan& pk, int D,
                                                   const uint32_t* tag, uint32_t epoch, int check_tags, uint32_t lane, PackedSums& o) {
  const uint32_t nw = pk.nw;
#pragma unroll
  for (int d = 0; d < 16; ++d) o.acc[d] = 0ull;
  o.pods = 0ull, o.zero_keys = 0u;
  const unsigned long long cnt_mask = pk.cnt_width >= 64 ? ~0ull : (1ull << pk.cnt_width) - 1ull;
  for (int b0 = (int)lane; b0 < n_slabs; b0 += 256) {
    bool on[4];
    unsigned long long w[4][4];
    uint32_t zk[4];
    // which of this lane's four slabs this launch wrote (namespace-ordered scans leave most (chunk, workgroup) slabs alone):
    // the four tags as one batch of loads, then the records of the live slabs only
#pragma unroll
    for (int u = 0; u < 4; ++u) on[u] = b0 + 64 * u < n_slabs;
    if (check_tags) {
      uint32_t tg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) tg[u] = tag[min(b0 + 64 * u, n_slabs - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) on[u] = on[u] && tg[u] == epoch;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      w[u][0] = w[u][1] = w[u][2] = w[u][3] = 0ull, zk[u] = 0u;
      if (on[u]) {
        const unsigned long long* q = (const unsigned long long*)(base + (size_t)(b0 + 64 * u) * pitch);
        w[u][0] = q[0], w[u][1] = q[nw > 1u ? 1 : 0], w[u][2] = q[nw > 2u ? 2 : 0], w[u][3] = q[nw > 3u ? 3 : 0];
        zk[u] = (uint32_t)q[nw];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!on[u] || w[u][0] == 0ull) continue;  // nobody of that workgroup matched this throttle
      o.pods += w[u][0] & cnt_mask;
      o.zero_keys |= zk[u];
#pragma unroll
      for (int d = 0; d < 16; ++d)
        if (d < D && pk.width[d]) {
          const uint32_t k = pk.word[d];
          const unsigned long long ww = k == 0u ? w[u][0] : k == 1u ? w[u][1] : k == 2u ? w[u][2] : w[u][3];
          const unsigned long long m = pk.width[d] >= 64 ? ~0ull : (1ull << pk.width[d]) - 1ull;
          o.acc[d] += (ww >> pk.pos[d]) & m;
        }
    }
  }
  o.pods = wave_sum64(o.pods);
  if (o.pods == 0ull) return;  // wave-uniform
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) o.zero_keys |= (uint32_t)__shfl_xor((int)o.zero_keys, s);
#pragma unroll
  for (int d = 0; d < 16; ++d)
    if (d < D && pk.width[d]) o.acc[d] = wave_sum64(o.acc[d]);
}

extern __shared__ __attribute__((aligned(16))) unsigned char kt_smem[];

}  // namespace kt
