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

// The packed slabs (PackPlan) of one chunk summed over the workgroups' slabs, one BLOCK of 16 waves per tile of
// kRecTileUnits 8-byte units of the chunk's row of records (whole records: 64 / units of them).
//   * lane = unit, wave = slab class: wave w reads the tile out of slabs w, w + 16, ... (at most 256 slabs: sixteen loads
//     per lane, issued as one batch) — every load instruction of a wave covers 512 contiguous bytes, the slab area is read
//     exactly once and fully coalesced (one wave per record with lane = slab, the first version of this reduction, moved
//     a cache line per 8-byte word: ~80 MB through L2 for 10 MB of slabs);
//   * the words are never taken apart per slab: every word is split into its top field (shifted down) and the two
//     classes of the fields below it (PackPlan::even — every field then has kPackHeadroomBits of zeros above it) and
//     summed whole; the unit behind the words is the OR of the key masks of pods that carry a key with the value 0;
//   * the sixteen waves meet in LDS: tot[unit][class].
// Slabs a namespace-ordered scan left alone (multi-chunk programs: most of them) are skipped by their tags, wave-uniformly.
// packed_field() then lets the lane of (record, dimension) cut its total out of tot — no loop over dimensions anywhere.
constexpr int kRecBlock = 1024, kRecWaves = kRecBlock / 64, kRecTileUnits = 64;
constexpr int kMaxSlabsPerRecord = 1 << kPackHeadroomBits;
static_assert(kMaxSlabsPerRecord == 16 * kRecWaves, "sixteen slabs per wave");
struct RecSumsLds {
  unsigned long long red[kRecWaves][kRecTileUnits][kPackClasses];
  unsigned long long tot[kRecTileUnits][kPackClasses];
};
// Three steps, so that a caller can place its own loads between them (vmcnt counts in order):
//   record_slabs_live   which of this wave's slabs this launch spilled (tag loads: multi-chunk programs only)
//   record_slabs_issue  the sixteen loads of this lane's unit.  row0: the tile's first byte in slab 0; n_units: units of
//                       the tile that exist (whole records)
//   block_record_sums   class sums, the meeting in LDS (two barriers: every thread of the block must call)
struct RecSlabLoads {
  uint32_t live;  // bit i: slab (wave + 16 i) was spilled by this launch
  unsigned long long v[16];
};
__device__ __forceinline__ void record_slabs_live(int n_slabs, const uint32_t* tag, uint32_t epoch, int check_tags, RecSlabLoads& sl) {
  const uint32_t lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = (int)(w + 16u * lane);
  bool ok = lane < 16u && s < n_slabs;
  if (ok && check_tags) ok = tag[s] == epoch;
  sl.live = (uint32_t)__ballot(ok);
}
__device__ __forceinline__ void record_slabs_issue(const unsigned char* row0, size_t pitch, uint32_t n_units, RecSlabLoads& sl) {
  const uint32_t lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mine = lane < n_units;
  const unsigned char* q = row0 + (size_t)w * pitch + (size_t)lane * 8u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    sl.v[i] = 0ull;
    if (mine && ((sl.live >> i) & 1u)) sl.v[i] = *(const unsigned long long*)(q + (size_t)(16 * i) * pitch);
  }
}
__device__ __forceinline__ void block_record_sums(const RecSlabLoads& sl, const PackPlan& pk, RecSumsLds& lds) {
  const uint32_t x = threadIdx.x, lane = x & 63u, w = __builtin_amdgcn_readfirstlane(x >> 6);
  const uint32_t units = pk.rec_bytes >> 3, nw = pk.nw;
  const uint32_t k = lane % units;  // what this unit is: word k of its record, the key-mask unit (k == nw), padding
  // this lane's word: mask of its even fields, of everything below its top field, position of the top field
  // (key-mask unit, padding: everything in class 0)
  unsigned long long ev = ~0ull, low = ~0ull;
  uint32_t tp = 0u;
#pragma unroll
  for (int j = 0; j < (int)kPackMaxWords; ++j) {
    const bool me = k == (uint32_t)j && (uint32_t)j < nw;
    ev = me ? pk.even[j] : ev;
    low = me ? (1ull << pk.top_pos[j]) - 1ull : low;
    tp = me ? pk.top_pos[j] : tp;
  }
  const bool word = k < nw;
  unsigned long long a = 0ull, b = 0ull, c = 0ull, o = 0ull;
#pragma unroll
  for (int i = 0; i < 16; ++i) a += sl.v[i] & ev & low, b += sl.v[i] & ~ev & low, c += sl.v[i] >> tp, o |= sl.v[i];
  lds.red[w][lane][0] = k == nw ? o : a;
  lds.red[w][lane][1] = b;
  lds.red[w][lane][2] = word ? c : 0ull;
  __syncthreads();
  if (x < (uint32_t)(kPackClasses * kRecTileUnits)) {
    const uint32_t u = x / kPackClasses, cl = x % kPackClasses;
    const bool is_or = u % units == nw;
    unsigned long long t = 0ull;
#pragma unroll
    for (int ww = 0; ww < kRecWaves; ++ww) {
      const unsigned long long r = lds.red[ww][u][cl];
      t = is_or ? (t | r) : t + r;
    }
    lds.tot[u][cl] = t;
  }
  __syncthreads();
}
// a total (in request units) of the record whose first unit is ub, for the lane that asks: desc = pk.desc[k] of
// dimension k (0: no field) or pk.cnt_desc (the pod count)
__device__ __forceinline__ unsigned long long packed_field(const RecSumsLds& lds, uint32_t ub, uint32_t desc) {
  const uint32_t sel = desc & 31u, pos = (desc >> 8) & 63u, wext = (desc >> 16) & 127u, shift = (desc >> 24) & 63u;
  const unsigned long long s = lds.tot[ub + (sel >> 2)][sel & 3u];
  const unsigned long long m = wext >= 64u ? ~0ull : (1ull << wext) - 1ull;
  return wext ? ((s >> pos) & m) << shift : 0ull;
}
__device__ __forceinline__ unsigned long long packed_pods(const RecSumsLds& lds, uint32_t ub, const PackPlan& pk) {
  return packed_field(lds, ub, pk.cnt_desc);
}
__device__ __forceinline__ uint32_t packed_zero_keys(const RecSumsLds& lds, uint32_t ub, const PackPlan& pk) { return (uint32_t)lds.tot[ub + pk.nw][0]; }
// pk.desc[k] for the lane of dimension k (a select chain over the 16 scalars: no indexed access to kernel arguments)
__device__ __forceinline__ uint32_t packed_desc_of(const PackPlan& pk, int k, int D) {
  uint32_t v = 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j) v = (j < D && k == j) ? pk.desc[j] : v;
  return v;
}

extern __shared__ __attribute__((aligned(16))) unsigned char kt_smem[];

}  // namespace kt
