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

extern __shared__ __attribute__((aligned(16))) unsigned char kt_smem[];

}  // namespace kt
