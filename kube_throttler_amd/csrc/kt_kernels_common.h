// kt_kernels_common.h — device helpers shared by the dense and indexed kernels (gfx950, wave64).
#pragma once
#include "kt_device.h"

namespace kt {

// Requirement evaluation against a pod's label registers.
//   In            : the pod carries one of the requirement's (key,value) pairs
//   NotIn         : it carries none of them (absent key included)
//   Exists        : it carries the key ; DoesNotExist: it does not
// (labels.Requirement.Matches of k8s.io/apimachinery v0.26.4, restated — SURVEY.md Appendix B.)
template <int LT, bool KEYS>
__device__ __forceinline__ bool term_match(const SelProgram& sp, uint32_t g, const uint32_t (&lp)[LT],
                                           const uint32_t (&lk)[LT]) {
  bool ok = true;
  const uint32_t r1 = sp.term_req_off[g + 1];
  for (uint32_t r = sp.term_req_off[g]; r < r1; ++r) {
    const uint8_t op = sp.req_op[r];
    bool sat;
    if (op <= kOpNotIn) {
      bool in = false;
      const uint32_t j1 = sp.req_val_off[r + 1];
      for (uint32_t j = sp.req_val_off[r]; j < j1; ++j) {
        const uint32_t v = sp.req_val[j];
#pragma unroll
        for (int l = 0; l < LT; ++l) in |= lp[l] == v;
      }
      sat = (op == kOpIn) ? in : !in;
    } else if (KEYS) {
      bool has = false;
      const uint32_t k = sp.req_key[r];
#pragma unroll
      for (int l = 0; l < LT; ++l) has |= lk[l] == k;
      sat = (op == kOpExists) ? has : !has;
    } else {
      sat = false;  // unreachable: KEYS is set whenever the program has Exists/DoesNotExist
    }
    ok &= sat;
  }
  return ok;
}

// CheckThrottledFor for one matched (pod, throttle): first hit wins
// (throttle_types.go:128-153 / clusterthrottle_types.go:30-55 through the CheckRec algebra).
template <int DT>
__device__ __forceinline__ uint32_t classify(const CheckRec<DT>* rec, const int64_t (&v)[DT], uint32_t nzmask) {
  const uint32_t f = rec->flags;
  bool exc = (f & kRecExceedsByCount) != 0;
  bool ins = (f & kRecInsufficientByCount) != 0;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    const bool nz = (nzmask >> d) & 1u;
    exc |= nz & (v[d] > rec->thr[d]);
    ins |= nz & (v[d] > rec->head[d]);
  }
  const bool act = ((f & kRecActiveByCount) != 0) | ((nzmask & rec->active_mask) != 0);
  return exc ? 4u : act ? 2u : ins ? 3u : 1u;
}

template <int DT, int LT, bool KEYS>
struct PodRegs {
  uint32_t ns, flags, nzmask;
  uint32_t lp[LT];
  uint32_t lk[LT];
  int64_t v[DT];
};

template <int DT, int LT, bool KEYS>
__device__ __forceinline__ void load_pod(const PodTable& pods, int64_t p, PodRegs<DT, LT, KEYS>& r, bool want_req) {
  r.ns = pods.ns[p];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    r.lp[l] = l < pods.L ? pods.lpair[(int64_t)l * pods.cap + p] : 0u;
    r.lk[l] = (KEYS && l < pods.L) ? pods.lkey[(int64_t)l * pods.cap + p] : 0u;
  }
  r.nzmask = 0;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    r.v[d] = (want_req && d < pods.D) ? pods.req[(int64_t)p * pods.D + d] : 0;
    r.nzmask |= (r.v[d] != 0 ? 1u : 0u) << d;
  }
}

__device__ __forceinline__ uint64_t pack_summary(uint32_t n_exc, uint32_t n_act, uint32_t n_ins, bool err) {
  if (err) return 2ull;
  const uint64_t verdict = (n_exc | n_act | n_ins) ? 1ull : 0ull;
  return verdict | (uint64_t)n_exc << 4 | (uint64_t)n_act << 24 | (uint64_t)n_ins << 44;
}

}  // namespace kt
