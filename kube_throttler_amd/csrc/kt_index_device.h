// kt_index_device.h — device-side pieces shared by the index-driven kernels (gfx950): LDS pointer types, the generic
// requirement walk for rare term shapes, the in-order walk of throttles with unconvertible selectors.
// The scan itself is kt_bitmap_scan.h.
#pragma once
#include <cstdlib>

#include "kt_index.h"
#include "kt_kernels_common.h"
#include "kt_launch.h"

namespace kt {

constexpr int kBlockIx = 1024;       // one workgroup per CU: 16 waves = 4 per SIMD
constexpr int kMaxLds = 160 * 1024;  // gfx950 LDS per CU / per workgroup
constexpr int kCUs = 256;

// Generic requirement walk for the rare term shapes the index cannot decide from a TermRec / TermX (flag kPostComplex:
// more than two extra requirements, or value sets larger than three).  The namespace side is already decided by the
// nsrows bitmap of the scan.
template <int LT, bool KEYS>
struct Matcher {
  const SelProgram& sp;
  const uint32_t (&lp)[LT];
  const uint32_t (&lk)[LT];
  __device__ __forceinline__ bool rare(uint32_t g) const { return term_match<LT, KEYS>(sp, g, lp, lk); }
};

// Explicit LDS (address space 3) pointer types: tables staged in LDS must be read with ds_read, not
// through generic/flat addressing (which costs 64-bit address math and the flat-memory latency).
#define KT_LDS __attribute__((address_space(3)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // plain vector types: loadable from any address space
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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
// Throttles with an unconvertible podSelector term: in-order walk, error when the bad term is reached
// before a match (same semantics as the dense kernels; t is wave-uniform).  Returns bit 0 = matched, bit 1 = error
// (by value: reference out-parameters ended up in scratch memory).
constexpr uint32_t kSlowMatched = 1u, kSlowError = 2u;
template <int LT, bool KEYS>
__device__ __forceinline__ uint32_t walk_slow(const SelProgram& sp, int t, const uint32_t* ns_row, bool lane_on,
                                              const uint32_t (&lp)[LT], const uint32_t (&lk)[LT]) {
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
    const bool mt = applies && term_match<LT, KEYS>(sp, g, lp, lk);
    res |= mt ? kSlowMatched : 0u;
    open &= !mt;
  }
  return res;
}

extern __shared__ __attribute__((aligned(16))) unsigned char kt_smem[];

}  // namespace kt
