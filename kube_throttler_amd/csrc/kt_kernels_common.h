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

typedef uint32_t kt_u32x4 __attribute__((ext_vector_type(4)));
typedef long long kt_i64x2 __attribute__((ext_vector_type(2)));

// A pod's label row (PodTable::lpair / lkey, stride LS in {4, 8, 16}) as LT >= LS registers.  Every 128-bit
// load is issued unconditionally from an address inside the row (chunks past the row re-read chunk 0 and are
// zeroed afterwards), so the loads of a tile go out back to back with no control flow between them.
template <int LT, bool KEYS>
__device__ __forceinline__ void load_labels(const uint32_t* lpair, const uint32_t* lkey, int LS, int64_t p,
                                            uint32_t (&lp)[LT], uint32_t (&lk)[LT]) {
  const kt_u32x4* a = (const kt_u32x4*)(lpair + p * LS);
  const kt_u32x4* k = (const kt_u32x4*)(lkey + p * LS);
  kt_u32x4 v[LT / 4], w[LT / 4];
#pragma unroll
  for (int q = 0; q < LT / 4; ++q) {
    const int qq = 4 * q < LS ? q : 0;
    v[q] = a[qq];
    if (KEYS) w[q] = k[qq];
  }
#pragma unroll
  for (int q = 0; q < LT / 4; ++q) {
    const uint32_t keep = 4 * q < LS ? ~0u : 0u;
    lp[4 * q] = v[q].x & keep, lp[4 * q + 1] = v[q].y & keep, lp[4 * q + 2] = v[q].z & keep, lp[4 * q + 3] = v[q].w & keep;
    if (KEYS) lk[4 * q] = w[q].x & keep, lk[4 * q + 1] = w[q].y & keep, lk[4 * q + 2] = w[q].z & keep, lk[4 * q + 3] = w[q].w & keep;
    else lk[4 * q] = lk[4 * q + 1] = lk[4 * q + 2] = lk[4 * q + 3] = 0u;
  }
}

// A pod's request row (stride DS, even) as DT/2 two-dimension pieces, same discipline.
template <int DT>
__device__ __forceinline__ void load_request_pieces(const int64_t* req, int DS, int64_t p, kt_i64x2 (&x)[DT / 2]) {
  const kt_i64x2* a = (const kt_i64x2*)(req + p * DS);
#pragma unroll
  for (int q = 0; q < DT / 2; ++q) x[q] = a[2 * q < DS ? q : 0];
#pragma unroll
  for (int q = 0; q < DT / 2; ++q) {
    const long long keep = 2 * q < DS ? -1ll : 0ll;
    x[q].x &= keep, x[q].y &= keep;
  }
}
template <int DT>
__device__ __forceinline__ void load_requests(const int64_t* req, int DS, int64_t p, int64_t (&v)[DT]) {
  kt_i64x2 x[DT / 2];
  load_request_pieces<DT>(req, DS, p, x);
#pragma unroll
  for (int q = 0; q < DT / 2; ++q) v[2 * q] = x[q].x, v[2 * q + 1] = x[q].y;
}

template <int DT, int LT, bool KEYS>
__device__ __forceinline__ void load_pod(const PodTable& pods, int64_t p, PodRegs<DT, LT, KEYS>& r, bool want_req) {
  r.ns = pods.ns[p];
  load_labels<LT, KEYS>(pods.lpair, pods.lkey, pods.LS, p, r.lp, r.lk);
  r.nzmask = 0;
  if (want_req) {
    load_requests<DT>(pods.req, pods.DS, p, r.v);
  } else {
#pragma unroll
    for (int d = 0; d < DT; ++d) r.v[d] = 0;
  }
#pragma unroll
  for (int d = 0; d < DT; ++d) r.nzmask |= (r.v[d] != 0 ? 1u : 0u) << d;
}

__device__ __forceinline__ uint64_t pack_summary(uint32_t n_exc, uint32_t n_act, uint32_t n_ins, bool err) {
  if (err) return 2ull;
  const uint64_t verdict = (n_exc | n_act | n_ins) ? 1ull : 0ull;
  return verdict | (uint64_t)n_exc << 4 | (uint64_t)n_act << 24 | (uint64_t)n_ins << 44;
}

}  // namespace kt
