// kt_kernels_admit.hip — sequential admission of a pod queue with reservation, gfx950.
//
// The scheduler admits pods one at a time: PreFilter(pod) (plugin.go:148-215) and, on Success, Reserve(pod)
// (plugin.go:217-239 -> [Cluster]ThrottleController.Reserve, throttle_controller.go:271-300 ->
// reservedResourceAmounts.addPod, reserved_resource_amounts.go:66-77), which adds ResourceAmountOfPod(pod) to the
// reserved amount of every throttle that affects the pod; the next pod's CheckThrottledFor sees it in steps 3
// and 4 (throttle_types.go:142-150).  The dependency chain is inherent, so this is ONE wave walking the queue in
// order with the whole mutable state (reserved amounts of all throttles) resident in LDS:
//
//   input   status matrix [n][T] of a preceding kt_check launch over the same queue: which throttles affect
//           which pod (selector side, evaluated for all pods in parallel there), error rows = 255
//   per pod (1) the row's nonzero bytes -> the pod's affected-throttle list (16 bytes per lane, ballot/mbcnt append)
//           (2) lane = (affected throttle, dimension): the four CheckThrottledFor steps against
//               threshold / status.used / status.throttled (HBM, read-only) and the CURRENT reserved row (LDS),
//               verdict bits OR-reduced over the lanes of a throttle
//           (3) summary word as PreFilter would return it at this point; the matrix row is rewritten with the
//               statuses the pod actually met
//           (4) verdict Success => reserved[t] += ResourceAmountOfPod(pod) for every affected throttle (LDS)
//   output  per-pod summary words, the rewritten matrix, and (commit) the reserved tables in HBM.
// When the state does not fit in LDS (thousands of throttles) it lives in an HBM scratch buffer instead (same code,
// L2 latency per step).
#include "kt_index_device.h"

namespace kt {

struct AdmitArgs {
  const uint32_t* pod_flags;
  const int64_t* req;
  const int64_t* rows;  // nullable: queue position -> pod table row
  int64_t n;
  ThrTables tt;
  unsigned char* scratch;  // state in HBM when it does not fit LDS (nullable)
  uint8_t* status;    // [n][T] in/out
  uint64_t* summary;  // [n] out
  int32_t T, D, DS, on_equal, commit;
  uint32_t off_rv, off_rc, off_rp, off_list, list_cap;
};

// the sums of used + reserved (+ the pod) are formed in 128 bits: an all-reduced `used` may come close to int64's end
__device__ __forceinline__ bool admit_cmp(__int128 a, int64_t b, bool eq) { return eq ? a >= (__int128)b : a > (__int128)b; }

// The mutable state (reserved amounts of all throttles) lives in LDS when it fits; otherwise in a scratch buffer in
// HBM that only this wave touches, read and written through L2 (agent-scope atomics: never served from a stale L1 line).
template <bool IN_LDS>
struct AdmitState;
template <>
struct AdmitState<true> {
  KT_LDS int64_t* rv;
  KT_LDS int64_t* rc;
  KT_LDS uint32_t* rp;
  __device__ __forceinline__ int64_t ld_v(int i) const { return rv[i]; }
  __device__ __forceinline__ void st_v(int i, int64_t x) const { rv[i] = x; }
  __device__ __forceinline__ int64_t ld_c(int i) const { return rc[i]; }
  __device__ __forceinline__ void st_c(int i, int64_t x) const { rc[i] = x; }
  __device__ __forceinline__ uint32_t ld_p(int i) const { return rp[i]; }
  __device__ __forceinline__ void st_p(int i, uint32_t x) const { rp[i] = x; }
};
template <>
struct AdmitState<false> {
  int64_t* rv;
  int64_t* rc;
  uint32_t* rp;
  __device__ __forceinline__ int64_t ld_v(int i) const { return __hip_atomic_load(rv + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ void st_v(int i, int64_t x) const { __hip_atomic_store(rv + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ int64_t ld_c(int i) const { return __hip_atomic_load(rc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ void st_c(int i, int64_t x) const { __hip_atomic_store(rc + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ uint32_t ld_p(int i) const { return __hip_atomic_load(rp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ void st_p(int i, uint32_t x) const { __hip_atomic_store(rp + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

template <int DT, bool IN_LDS>
__global__ __launch_bounds__(kWave) void kt_admit_sequential(const AdmitArgs a) {
  KT_LDS unsigned char* lds = (KT_LDS unsigned char*)kt_smem;
  AdmitState<IN_LDS> st;
  if constexpr (IN_LDS) {
    st.rv = (KT_LDS int64_t*)(lds + a.off_rv);    // [T][D] reserved requests
    st.rc = (KT_LDS int64_t*)(lds + a.off_rc);    // [T]    reserved pod count
    st.rp = (KT_LDS uint32_t*)(lds + a.off_rp);   // [T]    presence mask | has_count << 31
  } else {
    st.rv = (int64_t*)(a.scratch + a.off_rv);
    st.rc = (int64_t*)(a.scratch + a.off_rc);
    st.rp = (uint32_t*)(a.scratch + a.off_rp);
  }
  lds_u32wp list = (lds_u32wp)(lds + a.off_list);            // affected throttles of the current pod
  const int T = a.T, D = a.D;
  const uint32_t lane = threadIdx.x;
  const ThrTables& tt = a.tt;
  for (int t = (int)lane; t < T; t += kWave) {
    const uint32_t p = tt.reserved.present[t];
    for (int d = 0; d < D; ++d) st.st_v(t * D + d, ((p >> d) & 1u) ? tt.reserved.v[(size_t)t * D + d] : 0);
    const bool hc = tt.reserved.has_count[t] != 0;
    st.st_c(t, hc ? tt.reserved.count[t] : 0);
    st.st_p(t, p | (hc ? 0x80000000u : 0u));
  }
  if (!IN_LDS) __threadfence();
  constexpr int MPW = kWave / DT;
  const uint32_t d = lane % DT, ml = lane / DT;
  const bool d_in = (int)d < D;
  const bool eq = a.on_equal != 0;
  for (int64_t i = 0; i < a.n; ++i) {
    const int64_t p = a.rows ? a.rows[i] : i;
    const uint32_t fl = a.pod_flags[p];
    const uint32_t present = fl >> kPresentShift;
    uint8_t* row = a.status + i * T;
    // ---- (1) affected throttles: nonzero bytes of the matrix row, 16 per lane and chunk
    uint32_t n_aff = 0;  // wave-uniform
    bool err = false;
    for (int c0 = 0; c0 < T; c0 += kWave * 16) {
      const int b0 = c0 + (int)lane * 16;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (b0 < T) v = *(const u32x4*)(row + b0);  // the buffer has slack past the last row
      uint32_t nzm = 0;                            // bit k: byte k is nonzero
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t byte = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
        if (b0 + k < T && byte != 0) nzm |= 1u << k;
        err |= (b0 + k < T) && byte == 255u;
      }
      while (__ballot(nzm != 0) != 0ull) {
        const bool has = nzm != 0;
        const uint32_t k = (uint32_t)__ffs((int)nzm) - 1u;
        nzm &= nzm - 1u;
        const uint64_t mk = __ballot(has);
        const uint32_t pos = n_aff + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (has && pos < a.list_cap) list[pos] = (uint32_t)b0 + k;
        n_aff += (uint32_t)__popcll(mk);
      }
    }
    if (__ballot(err) != 0ull || !(fl & kPodValid) || n_aff > a.list_cap) {
      // error row (selector / namespace error, plugin.go:154-168), empty row, or a pod affected by more
      // throttles than the list holds: the pre-computed summary stands and nothing is reserved
      continue;
    }
    // ---- (2) lane = (affected throttle, dimension)
    uint32_t n_exc = 0, n_act = 0, n_ins = 0;
    const int64_t v = d_in ? a.req[p * a.DS + d] : 0;
    const bool nz = v != 0;
    for (uint32_t base = 0; base < n_aff; base += MPW) {
      const uint32_t j = base + ml;
      const bool vv = j < n_aff;
      const uint32_t t = list[vv ? j : 0u];
      const uint32_t tf = tt.flags[t];
      // threshold := status.calculatedThreshold if calculatedAt != zero else spec.threshold (throttle_types.go:129-132)
      const AmountTab& th = (tf & kThrCalcAtNonzero) ? tt.calc : tt.spec;
      const uint32_t th_p = th.present[t], u_p = tt.used.present[t], r_pw = st.ld_p(t);
      const bool eq3 = (tf & kThrCluster) ? eq : true;  // throttle_types.go:143 vs clusterthrottle_types.go:45
      uint32_t bits = 0;
      if (vv && d_in && ((th_p >> d) & 1u)) {
        const int64_t tv = th.v[(size_t)t * D + d];
        const int64_t uv = ((u_p >> d) & 1u) ? tt.used.v[(size_t)t * D + d] : 0;
        const int64_t rvd = st.ld_v(t * D + d);
        if (nz && v > tv) bits |= 1u;                                                        // step 1
        if (nz && (((u_p | r_pw) >> d) & 1u) && admit_cmp((__int128)uv + rvd, tv, eq3)) bits |= 2u;    // step 3
        if (nz && admit_cmp((__int128)uv + v + rvd, tv, eq)) bits |= 4u;                               // step 4
      }
      if (vv && nz && ((tt.thrl_flag[t] & tt.thrl_has[t]) >> d) & 1u) bits |= 2u;            // step 2
      if (vv && d == 0) {  // resourceCounts.pod
        const bool th_hc = th.has_count[t] != 0;
        const int64_t th_c = th.count[t];
        const bool u_hc = tt.used.has_count[t] != 0, r_hc = (r_pw >> 31) != 0;
        const int64_t u_c = u_hc ? tt.used.count[t] : 0, r_c = st.ld_c(t);
        if (th_hc && 1 > th_c) bits |= 1u;
        if ((tf & kThrThrottledPod) || (th_hc && (u_hc || r_hc) && admit_cmp((__int128)u_c + r_c, th_c, eq3))) bits |= 2u;
        if (th_hc && admit_cmp((__int128)u_c + 1 + r_c, th_c, eq)) bits |= 4u;
      }
#pragma unroll
      for (int o = DT / 2; o >= 1; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o);
      const uint32_t st = (bits & 1u) ? 4u : (bits & 2u) ? 2u : (bits & 4u) ? 3u : 1u;
      const bool lead = vv && d == 0;
      if (lead) row[t] = (uint8_t)st;
      n_exc += (uint32_t)__popcll(__ballot(lead && st == 4u));
      n_act += (uint32_t)__popcll(__ballot(lead && st == 2u));
      n_ins += (uint32_t)__popcll(__ballot(lead && st == 3u));
    }
    if (lane == 0) a.summary[i] = pack_summary(n_exc, n_act, n_ins, false);
    // ---- (4) Success: Reserve on every affected throttle
    if ((n_exc | n_act | n_ins) == 0) {
      for (uint32_t base = 0; base < n_aff; base += MPW) {
        const uint32_t j = base + ml;
        if (j < n_aff) {
          const uint32_t t = list[j];
          if (d_in && ((present >> d) & 1u)) st.st_v(t * D + d, st.ld_v(t * D + d) + v);
          if (d == 0) {
            st.st_c(t, st.ld_c(t) + 1);
            st.st_p(t, st.ld_p(t) | present | 0x80000000u);
          }
        }
      }
    }
  }
  if (a.commit) {
    for (int t = (int)lane; t < T; t += kWave) {
      const uint32_t w = st.ld_p(t);
      for (int dd = 0; dd < D; ++dd) tt.reserved.v[(size_t)t * D + dd] = st.ld_v(t * D + dd);
      tt.reserved.present[t] = w & 0x7FFFFFFFu;
      tt.reserved.has_count[t] = (uint8_t)(w >> 31);
      tt.reserved.count[t] = st.ld_c(t);
    }
  }
}

// LDS: state + list when the state fits, else the list alone
size_t admit_state_bytes(int T, int D) {
  return ((size_t)T * D * 8 + 15) / 16 * 16 + ((size_t)T * 8 + 15) / 16 * 16 + ((size_t)T * 4 + 15) / 16 * 16;
}

// scratch: admit_state_bytes(T, D) bytes of device memory, used when the state does not fit in LDS (or when forced)
bool launch_admit(const PodTable& pods, int64_t n, const int64_t* rows_dev, const ThrTables& tt, int T, int D,
                  bool on_equal, bool commit, uint8_t* status, uint64_t* summary, void* scratch, bool force_global,
                  hipStream_t s) {
  const size_t list_bytes = ((size_t)T * 4 + 15) / 16 * 16;
  const bool in_lds = !force_global && admit_state_bytes(T, D) + list_bytes <= (size_t)kMaxLds;
  if (!in_lds && (!scratch || list_bytes > (size_t)kMaxLds)) return false;
  AdmitArgs a{};
  a.pod_flags = pods.flags, a.req = pods.req, a.rows = rows_dev, a.n = n, a.tt = tt;
  a.status = status, a.summary = summary, a.scratch = (unsigned char*)scratch;
  a.T = T, a.D = D, a.DS = pods.DS, a.on_equal = on_equal ? 1 : 0, a.commit = commit ? 1 : 0;
  uint32_t o = 0;
  auto take = [&](size_t bytes) { uint32_t r = o; o += (uint32_t)((bytes + 15) & ~(size_t)15); return r; };
  a.off_rv = take((size_t)T * D * 8);  // offsets inside LDS or inside the scratch buffer
  a.off_rc = take((size_t)T * 8);
  a.off_rp = take((size_t)T * 4);
  a.list_cap = (uint32_t)T;  // a pod can be affected by every throttle
  if (!in_lds) o = 0;
  a.off_list = take((size_t)a.list_cap * 4);
  const int DT = dt_bucket(D);
  const size_t lds_bytes = o;
#define KT_ADMIT_CASE(DT_)                                                                                        \
  {                                                                                                              \
    auto kfn = in_lds ? kt_admit_sequential<DT_, true> : kt_admit_sequential<DT_, false>;                        \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);     \
    hipLaunchKernelGGL(kfn, dim3(1), dim3(kWave), lds_bytes, s, a);                                              \
  }
  if (DT == 4) KT_ADMIT_CASE(4) else if (DT == 8) KT_ADMIT_CASE(8) else KT_ADMIT_CASE(16)
#undef KT_ADMIT_CASE
  return true;
}

}  // namespace kt
